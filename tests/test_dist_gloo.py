"""world_size-2 gloo test (CPU): the N>1 path of bench.py -- one DB broadcast, query sharding, rank-order gather -- gives
the same per-query results as a single process.  The per-query compute is the oracle here (no GPU in this container);
what is under test is foldseek_amd/dist.py."""
import os
import sys
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, ws, port, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    from foldseek_amd import synth, dist as fdist
    import helpers
    q3, qa = synth.make_queries(5, seed=3, mean_len=80, lo=40, hi=120)
    db = synth.make_db(60, (q3, qa), seed=4, homologs_per_query=5, mean_len=80, lo=30, hi=150) if rank == 0 else None
    tensors, hdb = fdist.broadcast_db(db, torch.device("cpu"))
    assert tensors[0].numel() == hdb.data3di.size and tensors[3].numel() == hdb.n
    lo, hi = fdist.shard_range(len(q3), rank, ws)
    mine = []
    for qi in range(lo, hi):
        sc = helpers.o_ungapped_scores(q3[qi], hdb, True)
        sel = helpers.o_prefilter_select(sc, 30, -1, 10)
        mine.append((qi, sel["key"].tolist(), sel["score"].tolist()))
    allr = fdist.gather_objects(mine)
    t = fdist.max_over_ranks(float(rank + 1), torch.device("cpu"))
    assert t == float(ws)
    if rank == 0:
        flat = [x for part in allr for x in part]
        ret.put(flat)
    dist.destroy_process_group()


def test_broadcast_and_query_sharding_world2():
    sys.path.insert(0, ROOT)
    from foldseek_amd import synth, dist as fdist
    import helpers
    assert [fdist.shard_range(5, r, 2) for r in range(2)] == [(0, 3), (3, 5)]
    assert [fdist.shard_range(8, r, 3) for r in range(3)] == [(0, 3), (3, 6), (6, 8)]
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = 29500 + (os.getpid() % 500)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    flat = ret.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    q3, qa = synth.make_queries(5, seed=3, mean_len=80, lo=40, hi=120)
    db = synth.make_db(60, (q3, qa), seed=4, homologs_per_query=5, mean_len=80, lo=30, hi=150)
    assert [x[0] for x in flat] == list(range(5))
    for qi, keys, scores in flat:
        sel = helpers.o_prefilter_select(helpers.o_ungapped_scores(q3[qi], db, True), 30, -1, 10)
        assert keys == sel["key"].tolist() and scores == sel["score"].tolist()


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_gpus2_dry_run_starts_two_ranks(scaling):
    """`python bench.py --gpus 2` (no torchrun around it) must come up with two ranks through the BENCH code path: the
    self-launch, process-group init (gloo here, nccl on GPUs), vectorised DB generation on rank 0, one broadcast, query
    sharding.  --dry-run stops before the first device call."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run", "--targets", "2500", "--steps", "3",
                        "--warmup", "1", "--group", "4", "--scaling", scaling], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["dry_run"] is True and out["scaling"] == scaling and out["backend"] == "gloo"
    ranks = out["ranks"]
    assert [x["rank"] for x in ranks] == [0, 1]
    assert ranks[0]["db_digest"] == ranks[1]["db_digest"] and ranks[0]["db_entries"] == ranks[1]["db_entries"] == 2500
    timed = [x["timed_queries"] for x in ranks]
    assert timed == ([12, 12] if scaling == "weak" else [6, 6])     # weak: steps x group per rank; strong: the same 12 split


def test_bench_gpus8_dry_run_strong_scaling_shares():
    """the 8-rank shape of the driver's largest run, through the bench code path (gloo, no device work): every rank gets the one broadcast
    DB, the metric's 1k-query strong-scaling job splits into 8 contiguous shares of 128, and a rank with that few queries takes steps of 32"""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--dry-run", "--targets", "2000", "--steps", "16",
                        "--warmup", "1", "--group", "64", "--scaling", "strong"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 8 and out["dry_run"] is True and out["backend"] == "gloo"
    ranks = out["ranks"]
    assert [x["rank"] for x in ranks] == list(range(8))
    assert len({x["db_digest"] for x in ranks}) == 1 and all(x["db_entries"] == 2000 for x in ranks)
    assert [x["timed_queries"] for x in ranks] == [128] * 8 and sum(x["timed_queries"] for x in ranks) == 16 * 64
    assert out["queries_per_step_effective"] == 32
