mkdir -p gpurun_out/r3e
timeout 900 python -m pytest tests/test_kmer_gpu.py -x -q > gpurun_out/r3e/kmer_gpu.log 2>&1; echo "kmer_gpu rc=$?"; tail -5 gpurun_out/r3e/kmer_gpu.log
FSGPU_KMER_TRACE=1 timeout 600 python tools/kmer_bench.py 1000000 128 2 > gpurun_out/r3e/kb.log 2>&1; echo rc=$?; grep -v "^kmer batch" gpurun_out/r3e/kb.log | tail -4; grep "^kmer batch" gpurun_out/r3e/kb.log | tail -4
FSGPU_KMER_TRACE=1 timeout 600 python bench.py --type2-steps 0 --allvsall-steps 0 --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/r3e/bench.json 2> gpurun_out/r3e/bench.err; echo "bench rc=$?"; grep "^kmer batch" gpurun_out/r3e/bench.err | tail -8
