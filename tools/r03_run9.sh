mkdir -p gpurun_out/r3i
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3i/gpu_all.log 2>&1; echo "gpu_all rc=$?"; tail -6 gpurun_out/r3i/gpu_all.log
timeout 600 python bench.py > gpurun_out/r3i/bench.json 2> gpurun_out/r3i/bench.err; echo "bench rc=$?"; tail -3 gpurun_out/r3i/bench.err
timeout 600 python tools/kmer_fuzz.py 40 303 > gpurun_out/r3i/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -3 gpurun_out/r3i/fuzz.log
