#!/bin/bash
# Round-2 evidence run (one gpurun call): default bench line, rocprofv3 kernel trace of the same command, separate --pmc
# passes (counters never combined with tracing, MI355X_MICROARCH.md) over a short bench run.  Output: gpurun_out/r02_prof/.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02_prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
TAG=${1:-x}
python $R/bench.py > $O/${TAG}_bench_n1.json 2> $O/${TAG}_bench_n1.err
python $R/bench.py --targets 100000 > $O/${TAG}_bench_n1_100k_targets.json 2>> $O/${TAG}_bench_n1.err
python $R/bench.py --alignment-type 2 --no-kmer --no-cpu-baseline > $O/${TAG}_bench_n1_alntype2.json 2>> $O/${TAG}_bench_n1.err
rm -rf /tmp/kt && rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $R/bench.py --no-cpu-baseline > /tmp/kt.log 2>&1
python $R/tools/rocprof_summary.py $(find /tmp/kt -name "*.db" | head -1) > $O/${TAG}_kernel_trace_bench_default.txt 2>&1
SHORT="--steps 3 --warmup 1 --no-cpu-baseline --kmer-queries 64"
pass() { rm -rf /tmp/pmc_$1; rocprofv3 --pmc "$@" -d /tmp/pmc_$1 -o p --output-format csv -- python $R/bench.py $SHORT > /tmp/pmc_$1.log 2>&1; }
pass SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY
pass SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
pass FETCH_SIZE
pass WRITE_SIZE
python $R/tools/pmc_family.py /tmp/pmc_SQ_WAVES /tmp/pmc_SQ_LDS_BANK_CONFLICT /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE --json $O/${TAG}_pmc_bench_1M.json > $O/${TAG}_pmc_bench_1M.txt 2>&1
grep -h "^{" /tmp/pmc_FETCH_SIZE.log | tail -1 > $O/${TAG}_pmc_bench_1M_benchline.json
