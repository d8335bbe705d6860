mkdir -p gpurun_out/r3c
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3c/gpu_all.log 2>&1; echo "gpu_all rc=$?"; tail -8 gpurun_out/r3c/gpu_all.log
timeout 600 python bench.py --type2-steps 0 > gpurun_out/r3c/bench.json 2> gpurun_out/r3c/bench.err; echo "bench rc=$?"; tail -3 gpurun_out/r3c/bench.err
timeout 600 python tools/kmer_fuzz.py --rounds 40 --seed 303 > gpurun_out/r3c/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -3 gpurun_out/r3c/fuzz.log
