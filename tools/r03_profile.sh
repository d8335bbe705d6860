#!/bin/bash
# Round-3 evidence, two gpurun calls (a bench line can only quote PMC traffic that was collected BEFORE it and converted by
# tools/pmc_to_traffic.py, which stamps it with the hash of the kernel sources):
#   r03_profile.sh pmc TAG    separate --pmc passes (counters never combined with tracing, MI355X_MICROARCH.md) over a short bench run
#                             (k_gapless, k_sw2) and over ONE k-mer prefilter batch of 32 queries at 1M targets
#   r03_profile.sh bench TAG  default bench line, the same with the DB's full query length range, rocprofv3 kernel traces of the
#                             default command and of three k-mer batches
# Output: gpurun_out/r03_prof/.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
MODE=${1:-bench}
TAG=${2:-x}
if [ "$MODE" = pmc ]; then
python $R/tools/csrc_hash.py k_gapless.hpp fs_kernels.h > $O/${TAG}_csrc_hash_gapless.txt
python $R/tools/csrc_hash.py k_kmer.hpp fsgpu_kmer.hip fs_kernels.h > $O/${TAG}_csrc_hash_kmer.txt
# --- main path counters (k_gapless, k_sw2): 3 steps of the default workload, no other legs
SHORT="--steps 3 --warmup 1 --no-cpu-baseline --no-kmer --type2-steps 0 --allvsall-steps 0"
pass() { rm -rf /tmp/pmc_$1; rocprofv3 --pmc "$@" -d /tmp/pmc_$1 -o p --output-format csv -- python $R/bench.py $SHORT > /tmp/pmc_$1.log 2>&1; }
pass SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY
pass SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
pass FETCH_SIZE
pass WRITE_SIZE
python $R/tools/pmc_family.py /tmp/pmc_SQ_WAVES /tmp/pmc_SQ_LDS_BANK_CONFLICT /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE --json $O/${TAG}_pmc_bench_1M.json > $O/${TAG}_pmc_bench_1M_steps3.txt 2>&1
grep -h "^{" /tmp/pmc_SQ_WAVES.log | tail -1 > $O/${TAG}_pmc_bench_1M_benchline.json
# --- k-mer prefilter counters: ONE batch of 32 queries at 1M targets (tools/kmer_bench.py prints the batch's probe / hit counts);
#     dispatches before the first k_kmer_count belong to the index build and are dropped
kpass() { rm -rf /tmp/kpmc_$1; rocprofv3 --pmc "$@" -d /tmp/kpmc_$1 -o p --output-format csv -- python $R/tools/kmer_bench.py 1000000 32 1 > /tmp/kpmc_$1.log 2>&1; }
kpass SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY
kpass SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
kpass FETCH_SIZE
kpass WRITE_SIZE
python $R/tools/pmc_family.py /tmp/kpmc_SQ_WAVES /tmp/kpmc_SQ_LDS_BANK_CONFLICT /tmp/kpmc_FETCH_SIZE /tmp/kpmc_WRITE_SIZE --from-first k_kmer_count --json $O/${TAG}_pmc_kmer_1M.json > $O/${TAG}_pmc_kmer_batch32_1M.txt 2>&1
grep -h "^COUNTS\|^rep\|^segments" /tmp/kpmc_FETCH_SIZE.log > $O/${TAG}_pmc_kmer_1M_counts.txt
else
python $R/bench.py > $O/${TAG}_bench_n1.json 2> $O/${TAG}_bench_n1.err
python $R/bench.py --query-len 30,2000 --no-kmer --type2-steps 0 --allvsall-steps 0 --no-cpu-baseline > $O/${TAG}_bench_n1_querylen_30_2000.json 2>> $O/${TAG}_bench_n1.err
rm -rf /tmp/kt && rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $R/bench.py --no-cpu-baseline > /tmp/kt.log 2>&1
python $R/tools/rocprof_summary.py $(find /tmp/kt -name "*.db" | head -1) > $O/${TAG}_kernel_trace_bench_default.txt 2>&1
grep -h "^{" /tmp/kt.log | tail -1 > $O/${TAG}_kernel_trace_benchline.json
rm -rf /tmp/kkt && rocprofv3 --kernel-trace --stats -d /tmp/kkt -o kt -- python $R/tools/kmer_bench.py 1000000 32 3 > /tmp/kkt.log 2>&1
python $R/tools/rocprof_summary.py $(find /tmp/kkt -name "*.db" | head -1) > $O/${TAG}_kernel_trace_kmer_batch32_1M.txt 2>&1
grep -h "^COUNTS\|^rep\|^segments" /tmp/kkt.log >> $O/${TAG}_kernel_trace_kmer_batch32_1M.txt
fi
ls -la $O | tail -20
