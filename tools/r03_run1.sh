mkdir -p gpurun_out/r3a
timeout 900 python -m pytest tests/test_hotpath_fullsize_gpu.py -x -q -k "1M" > gpurun_out/r3a/test1m.log 2>&1; echo "test1m rc=$?"
( time timeout 900 python bench.py ) > gpurun_out/r3a/bench.json 2> gpurun_out/r3a/bench.err; echo "bench rc=$?"
tail -3 gpurun_out/r3a/bench.err
timeout 600 python bench.py --emulate-rank-share 8 --no-kmer --no-cpu-baseline --type2-steps 0 --allvsall-steps 0 > gpurun_out/r3a/emu8_weak.json 2> gpurun_out/r3a/emu8_weak.err; echo "emu weak rc=$?"
timeout 600 python bench.py --emulate-rank-share 8 --scaling strong --steps 16 --no-kmer --no-cpu-baseline --type2-steps 0 --allvsall-steps 0 > gpurun_out/r3a/emu8_strong.json 2> gpurun_out/r3a/emu8_strong.err; echo "emu strong rc=$?"
tail -5 gpurun_out/r3a/test1m.log
