mkdir -p gpurun_out/r3d
FSGPU_KMER_TRACE=1 timeout 600 python tools/kmer_bench.py 1000000 128 2 > gpurun_out/r3d/kb.log 2>&1; echo rc=$?; tail -14 gpurun_out/r3d/kb.log
