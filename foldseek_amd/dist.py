"""Multi-GPU plumbing for the hot path: the target DB is broadcast ONCE (RCCL over xGMI when the tensors live in HBM,
gloo in the CPU tests) and queries are sharded across ranks; there is no per-step communication because every query's
result depends only on that query and the replicated, read-only DB."""
import numpy as np
import torch
import torch.distributed as dist

from .synth import PaddedDB


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_range(n_items, rank, world_size):
    """contiguous block of queries for this rank (sizes differ by at most one)"""
    base, rem = divmod(n_items, world_size)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def broadcast_db(db, device, src=0):
    """Rank `src` passes a PaddedDB, the others None.  Returns (tensors, host PaddedDB) on every rank; the four tensors
    (data3di, dataaa, offsets int64, lengths int32) live on `device` and can be handed to fsgpu_db_adopt_device."""
    rank, ws = world()
    if ws == 1:
        t = [torch.from_numpy(np.ascontiguousarray(db.data3di)).to(device), torch.from_numpy(np.ascontiguousarray(db.dataaa)).to(device),
             torch.from_numpy(np.ascontiguousarray(db.offsets, np.int64)).to(device), torch.from_numpy(np.ascontiguousarray(db.lengths, np.int32)).to(device)]
        return t, db
    meta = torch.zeros(2, dtype=torch.int64, device=device)
    if rank == src:
        meta = torch.tensor([db.n, db.data3di.size], dtype=torch.int64, device=device)
    dist.broadcast(meta, src)
    n, nbytes = int(meta[0]), int(meta[1])
    if rank == src:
        t = [torch.from_numpy(np.ascontiguousarray(db.data3di)).to(device), torch.from_numpy(np.ascontiguousarray(db.dataaa)).to(device),
             torch.from_numpy(np.ascontiguousarray(db.offsets, np.int64)).to(device), torch.from_numpy(np.ascontiguousarray(db.lengths, np.int32)).to(device)]
    else:
        t = [torch.empty(nbytes, dtype=torch.uint8, device=device), torch.empty(nbytes, dtype=torch.uint8, device=device),
             torch.empty(n + 1, dtype=torch.int64, device=device), torch.empty(n, dtype=torch.int32, device=device)]
    for x in t:
        dist.broadcast(x, src)
    if rank != src:
        db = PaddedDB(t[0].cpu().numpy(), t[1].cpu().numpy(), t[2].cpu().numpy(), t[3].cpu().numpy())
    return t, db


def gather_objects(obj, dst=0):
    """per-query results are independent: rank `dst` just concatenates the shards in rank order"""
    rank, ws = world()
    if ws == 1:
        return [obj]
    out = [None] * ws if rank == dst else None
    dist.gather_object(obj, out, dst=dst)
    return out


def max_over_ranks(x, device):
    rank, ws = world()
    if ws == 1:
        return x
    t = torch.tensor([x], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t[0])
