mkdir -p gpurun_out/r3b
timeout 900 python -m pytest tests/test_kmer_gpu.py -x -q > gpurun_out/r3b/kmer_gpu.log 2>&1; echo "kmer_gpu rc=$?"; tail -15 gpurun_out/r3b/kmer_gpu.log
