mkdir -p gpurun_out/r3k
timeout 1200 python -m pytest tests/test_reference_adapters_gpu.py tests/test_marv_dropin.py -x -q > gpurun_out/r3k/adapters.log 2>&1; echo "adapters rc=$?"; tail -30 gpurun_out/r3k/adapters.log
