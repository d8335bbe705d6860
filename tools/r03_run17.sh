mkdir -p gpurun_out/r3v
cd $GRAFT_REPO_ROOT
run() { tag=$1; shift; env "$@" FSGPU_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $PORT bench.py --gpus 2 --steps 6 --warmup 2 --targets 100000 --allvsall-targets 20000 --allvsall-steps 8 $EXTRA > gpurun_out/r3v/$tag.json 2> gpurun_out/r3v/$tag.err; echo "$tag rc=$?"; grep -i "fault\|Traceback\|rank.*line" gpurun_out/r3v/$tag.err | head -6; }
PORT=29521 EXTRA="" run plain X=1
PORT=29522 EXTRA="--no-kmer" run nokmer X=1
PORT=29523 EXTRA="--type2-steps 0" run notype2 X=1
PORT=29524 EXTRA="--no-kmer --type2-steps 0" run onlymain X=1
PORT=29525 EXTRA="" run serial AMD_SERIALIZE_KERNEL=3 HIP_LAUNCH_BLOCKING=1
