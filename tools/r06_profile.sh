#!/bin/bash
# Round-6 evidence (counters never combined with tracing, MI355X_MICROARCH.md):
#   r06_profile.sh emu TAG    bench.py --emulate-rank-share 8, weak and strong scaling: what ONE rank of an 8-rank node runs, on one GPU with that rank's 2 cores
#   r06_profile.sh pmc TAG    separate --pmc passes over a short bench run (k_gapless, k_sw3), over ONE k-mer prefilter batch of 32 queries at 1M
#                             targets, and over a short all-vs-all run (configs[4], 200k family DB)
#   r06_profile.sh bench TAG  default bench line, rocprofv3 kernel traces of the default command and of three k-mer batches
# Output: gpurun_out/r06_prof/.  tools/pmc_to_traffic.py turns the pmc files into profiles/pmc_traffic*.json.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
MODE=${1:-bench}
TAG=${2:-x}
#   r06_profile.sh pmck TAG   the k-mer and all-vs-all passes of `pmc` alone (the scan kernel's sources did not change: its pass of tag PREV=... is reused)
#   r06_profile.sh sw TAG     the batch SW alone: tools/sw2_probe.py for 16 .. 256 queries x 1000 targets without / with the 16-lane shape
if [ "$MODE" = sw ]; then
for n in 16 32 64 128 256; do for v in "FSGPU_SW3_MID=0" "FSGPU_SW3_MID=512"; do echo "== $n queries, $v"; env $v python $R/tools/sw2_probe.py $n 2>&1 | grep "alignment-type"; done; done > $O/${TAG}_sw_probe_batch_scaling.txt
elif [ "$MODE" = emu ]; then
python $R/bench.py --emulate-rank-share 8 --scaling weak > $O/${TAG}_bench_emulate_rank_share8_weak.json 2> $O/${TAG}_emu_weak.err
python $R/bench.py --emulate-rank-share 8 --scaling strong --steps 20 > $O/${TAG}_bench_emulate_rank_share8_strong.json 2> $O/${TAG}_emu_strong.err
elif [ "$MODE" = pmc ] || [ "$MODE" = pmck ]; then
python $R/tools/csrc_hash.py k_gapless.hpp fs_kernels.h > $O/${TAG}_csrc_hash_gapless.txt
python $R/tools/csrc_hash.py k_kmer.hpp fsgpu_kmer.hip fs_kernels.h > $O/${TAG}_csrc_hash_kmer.txt
python $R/tools/csrc_hash.py k_sw3.hpp k_sw.hpp fsgpu_sw3.hip fs_kernels.h > $O/${TAG}_csrc_hash_sw.txt
if [ "$MODE" = pmc ]; then
SHORT="--steps 3 --warmup 1 --no-cpu-baseline --no-kmer --type2-steps 0 --allvsall-steps 0 --fullrange-steps 0 --single-targets 0"
pass() { rm -rf /tmp/pmc_$1; rocprofv3 --pmc "$@" -d /tmp/pmc_$1 -o p --output-format csv -- python $R/bench.py $SHORT > /tmp/pmc_$1.log 2>&1; }
pass SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY
pass SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
pass FETCH_SIZE
pass WRITE_SIZE
python $R/tools/pmc_family.py /tmp/pmc_SQ_WAVES /tmp/pmc_SQ_LDS_BANK_CONFLICT /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE --json $O/${TAG}_pmc_bench_1M.json > $O/${TAG}_pmc_bench_1M_steps3.txt 2>&1
grep -h "^{" /tmp/pmc_SQ_WAVES.log | tail -1 > $O/${TAG}_pmc_bench_1M_benchline.json
# the same short run with --alignment-type 2 (configs[3]): FETCH / WRITE of k_sw3 with the AA table -> align_type2's align_roofline.traffic
pass2() { rm -rf /tmp/pmc2_$1; rocprofv3 --pmc "$@" -d /tmp/pmc2_$1 -o p --output-format csv -- python $R/bench.py $SHORT --alignment-type 2 > /tmp/pmc2_$1.log 2>&1; }
pass2 FETCH_SIZE
pass2 WRITE_SIZE
python $R/tools/pmc_family.py /tmp/pmc2_FETCH_SIZE /tmp/pmc2_WRITE_SIZE --json $O/${TAG}_pmc_bench_1M_t2.json > $O/${TAG}_pmc_bench_1M_t2_steps3.txt 2>&1
grep -h "^{" /tmp/pmc2_FETCH_SIZE.log | tail -1 > $O/${TAG}_pmc_bench_1M_t2_benchline.json
else
cp $R/profiles/${PREV:-r04_z}_pmc_bench_1M.json $O/${TAG}_pmc_bench_1M.json
fi
kpass() { rm -rf /tmp/kpmc_$1; rocprofv3 --pmc "$@" -d /tmp/kpmc_$1 -o p --output-format csv -- python $R/tools/kmer_bench.py 1000000 32 1 > /tmp/kpmc_$1.log 2>&1; }
kpass SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY
kpass SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
kpass FETCH_SIZE
kpass WRITE_SIZE
python $R/tools/pmc_family.py /tmp/kpmc_SQ_WAVES /tmp/kpmc_SQ_LDS_BANK_CONFLICT /tmp/kpmc_FETCH_SIZE /tmp/kpmc_WRITE_SIZE --from-first k_kmer_count --json $O/${TAG}_pmc_kmer_1M.json > $O/${TAG}_pmc_kmer_batch32_1M.txt 2>&1
grep -h "^COUNTS\|^rep\|^segments" /tmp/kpmc_FETCH_SIZE.log > $O/${TAG}_pmc_kmer_1M_counts.txt
# all-vs-all: 8 batches of 256 DB entries against the 200k family DB (k-mer prefilter + SW), no module run, no CPU baseline
AV="--workload allvsall --targets 200000 --steps 8 --warmup 2 --no-cpu-baseline --kmer-threads 1 --allvsall-batch 1024"
apass() { rm -rf /tmp/apmc_$1; rocprofv3 --pmc "$@" -d /tmp/apmc_$1 -o p --output-format csv -- python $R/bench.py $AV > /tmp/apmc_$1.log 2>&1; }
apass FETCH_SIZE
apass WRITE_SIZE
python $R/tools/pmc_family.py /tmp/apmc_FETCH_SIZE /tmp/apmc_WRITE_SIZE --from-first k_kmer_count --json $O/${TAG}_pmc_allvsall_200k.json > $O/${TAG}_pmc_allvsall_200k.txt 2>&1
grep -h "^{" /tmp/apmc_FETCH_SIZE.log | tail -1 > $O/${TAG}_pmc_allvsall_200k_benchline.json
else
python $R/bench.py > $O/${TAG}_bench_n1.json 2> $O/${TAG}_bench_n1.err
rm -rf /tmp/kt && rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $R/bench.py --no-cpu-baseline > /tmp/kt.log 2>&1
python $R/tools/rocprof_summary.py $(find /tmp/kt -name "*.db" | head -1) > $O/${TAG}_kernel_trace_bench_default.txt 2>&1
grep -h "^{" /tmp/kt.log | tail -1 > $O/${TAG}_kernel_trace_benchline.json
rm -rf /tmp/kkt && rocprofv3 --kernel-trace --stats -d /tmp/kkt -o kt -- python $R/tools/kmer_bench.py 1000000 32 3 > /tmp/kkt.log 2>&1
python $R/tools/rocprof_summary.py $(find /tmp/kkt -name "*.db" | head -1) > $O/${TAG}_kernel_trace_kmer_batch32_1M.txt 2>&1
grep -h "^COUNTS\|^rep\|^segments" /tmp/kkt.log >> $O/${TAG}_kernel_trace_kmer_batch32_1M.txt
fi
ls -la $O | tail -20
