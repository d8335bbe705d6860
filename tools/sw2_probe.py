"""k_sw2 alone on the device: one batch of 32 queries x 1000 random targets (100k-target synthetic DB), forward pass, 3Di and 3Di+AA.
FSGPU_SW2_PAIRS=8|16|32 overrides the pairs per workgroup.  Prints device ms of the pass (HIP events) and Tcell/s."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from foldseek_amd import api, synth
q3, qa = synth.make_queries(64, seed=5000, lo=250, hi=450)
db = synth.make_db(100000, synth.make_queries(8, seed=1000, lo=250, hi=450))
ctx = api.Context(0); ctx.load_db(db)
rng = np.random.default_rng(1)
hits = [rng.choice(db.n, 1000, replace=False).astype(np.uint32) for _ in range(64)]
for at in (0, 2):
    par = api.default_params(); par.alignmentType = at
    s = api.Search(ctx, par)
    best = None
    for rep in range(5):
        r = s.align_batch(qa[:32], q3[:32], hits[:32])
        p = ctx.sw_last_passes()
        if best is None or p[0][0] < best[0][0]:
            best = p
    ms, cells = float(best[0][0]), float(best[0][1])
    print("alignment-type %d pairs/wg %s: forward pass %.3f ms per batch of 32 (%.4f ms/query), %.3e cells -> %.3f Tcell/s; records %d" % (
        at, os.environ.get("FSGPU_SW2_PAIRS", "default"), ms, ms / 32, cells, cells / ms / 1e9, sum(len(x) for x in r)), flush=True)
    s.close()
