mkdir -p gpurun_out/r3w
cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5 6; do
FSGPU_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29530+i)) bench.py --gpus 2 --steps 6 --warmup 2 --targets 100000 --allvsall-targets 20000 --allvsall-steps 8 > gpurun_out/r3w/run$i.json 2> gpurun_out/r3w/run$i.err; echo "run $i rc=$?"; grep -i "fault" gpurun_out/r3w/run$i.err | head -2
done
python -c "
import json
d=json.loads([l for l in open('gpurun_out/r3w/run6.json') if l.startswith('{')][-1])
print(d['n_gpus'], d['value'], d['alignments_per_query'], d['broadcast_backend'], d['kmer_prefilter']['queries_per_s'], d['allvsall']['queries_per_s'], d['allvsall']['n_gpus'], d['align_type2']['ms_per_query'])"
