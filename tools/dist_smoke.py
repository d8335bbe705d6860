"""One-GPU smoke of the collective calls bench.py makes at N > 1 (RCCL: init, broadcast of uint8/int64/int32 tensors,
all_reduce MAX on float64, barrier, gather_object) -- run as
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 tools/dist_smoke.py
The multi-rank branch of foldseek_amd.dist.broadcast_db is forced by reporting a world size of 2 to it (rank 0 is the
source, so no peer is needed); the library then adopts the broadcast tensors and runs one query."""
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from foldseek_amd import api, synth, dist as fdist

local_rank = int(os.environ.get("LOCAL_RANK", "0"))
dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
torch.cuda.set_device(local_rank)
dev = torch.device("cuda", local_rank)
q3, qa = synth.make_queries(4, seed=3, lo=100, hi=300)
db = synth.make_db(3000, (q3, qa), seed=4, homologs_per_query=20)
real_world = fdist.world
fdist.world = lambda: (0, 2)
tensors, db2 = fdist.broadcast_db(db, dev)
fdist.world = real_world
assert db2 is db and tensors[0].is_cuda and int(tensors[2][-1]) == int(db.offsets[-1])
t = torch.tensor([1.25], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
out = [None]
dist.gather_object({"rank": 0}, out, dst=0)
ctx = api.Context(local_rank)
ctx.adopt_device_db(tensors[0].data_ptr(), tensors[1].data_ptr(), tensors[2].data_ptr(), tensors[3].data_ptr(), db.n, db.data3di.size)
ctx._keep = (np.ascontiguousarray(db.data3di), np.ascontiguousarray(db.dataaa), np.ascontiguousarray(db.offsets, np.uint64),
             np.ascontiguousarray(db.lengths, np.int32))          # host copies for the backtrace stage, as bench.py keeps them
s = api.Search(ctx, api.default_params())
hits = s.prefilter(q3[0])
res = s.align(qa[0], q3[0], hits["id"])
print("dist smoke ok: nccl broadcast + adopt + search,", len(hits), "hits", float(t[0]), out)
s.close(); ctx.close()
dist.destroy_process_group()
