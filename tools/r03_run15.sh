mkdir -p gpurun_out/r3r
timeout 900 python -m pytest tests/test_kmer_gpu.py tests/test_kmer_golden.py tests/test_kmer_fullsize_gpu.py -x -q > gpurun_out/r3r/kmer_gpu.log 2>&1; echo "kmer_gpu rc=$?"; tail -5 gpurun_out/r3r/kmer_gpu.log
(timeout 600 python tools/kmer_fuzz.py 80 4242 2>&1 | tail -2) | tee gpurun_out/r3r/kmer_fuzz.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt && rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $GRAFT_REPO_ROOT/tools/kmer_bench.py 1000000 32 3 > /tmp/kt.log 2>&1
grep "^rep" /tmp/kt.log
python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $(find /tmp/kt -name "*.db" | head -1) > $GRAFT_REPO_ROOT/gpurun_out/r3r/kt_kmer_bench_1M.txt 2>&1
grep -E "k_kmer_(dup|emit|scatter|bincount|score)" $GRAFT_REPO_ROOT/gpurun_out/r3r/kt_kmer_bench_1M.txt | cut -c1-60,73-200
