mkdir -p gpurun_out/r3j
timeout 900 python -m pytest tests/test_modules_gpu.py tests/test_modules_vs_reference_binary.py -x -q > gpurun_out/r3j/modules.log 2>&1; echo "modules rc=$?"; tail -3 gpurun_out/r3j/modules.log
for th in 4 16; do timeout 600 python tools/allvsall_modules.py 200000 $th 20000 > gpurun_out/r3j/allvsall_modules_t$th.txt 2>&1; echo "allvsall t=$th rc=$?"; tail -2 gpurun_out/r3j/allvsall_modules_t$th.txt; done
