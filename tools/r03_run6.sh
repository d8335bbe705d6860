mkdir -p gpurun_out/r3f
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt && rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $GRAFT_REPO_ROOT/tools/kmer_bench.py 1000000 64 2 > /tmp/kt.log 2>&1
tail -3 /tmp/kt.log
python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $(find /tmp/kt -name "*.db" | head -1) > $GRAFT_REPO_ROOT/gpurun_out/r3f/kt_kmer_bench_1M.txt 2>&1
head -40 $GRAFT_REPO_ROOT/gpurun_out/r3f/kt_kmer_bench_1M.txt
