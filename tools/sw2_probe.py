"""The batch SW kernel alone on the device: one batch of 32 queries x 1000 random targets (100k-target synthetic DB), forward pass, 3Di and 3Di+AA.
Default: k_sw3 (compact queries, fsgpu_sw_multi_dir_c; FSGPU_SW3_WAVES=2|4|8 overrides the waves per workgroup); FSGPU_SW_PROFILES=1: k_sw2
(fsgpu_sw_multi_dir; FSGPU_SW2_PAIRS=8|16|32 overrides the pairs per workgroup).  Prints device ms of the pass (HIP events), Tcell/s, and the
fraction of the packed-VALU issue bound (1024 SIMDs x 128 cells / (14 [with AA: 15 for k_sw3, 16 for k_sw2] instructions x 4.3 cycles) at 2.4 GHz)."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from foldseek_amd import api, synth
NQ = int(sys.argv[1]) if len(sys.argv) > 1 else 32          # queries per batch
NP = int(sys.argv[2]) if len(sys.argv) > 2 else 1000        # pairs per query (8: the all-vs-all shape)
q3, qa = synth.make_queries(max(64, NQ), seed=5000, lo=250, hi=450)
db = synth.make_db(100000, synth.make_queries(8, seed=1000, lo=250, hi=450))
ctx = api.Context(0); ctx.load_db(db)
rng = np.random.default_rng(1)
hits = [rng.choice(db.n, NP, replace=False).astype(np.uint32) for _ in range(max(64, NQ))]
for at in (0, 2):
    par = api.default_params(); par.alignmentType = at
    s = api.Search(ctx, par)
    best = None
    for rep in range(5):
        r = s.align_batch(qa[:NQ], q3[:NQ], hits[:NQ])
        p = ctx.sw_last_passes()
        if best is None or p[0][0] < best[0][0]:
            best = p
    ms, cells = float(best[0][0]), float(best[0][1])
    legacy = os.environ.get("FSGPU_SW_PROFILES", "0") not in ("", "0")
    peak = 1024 * 128 / (((16 if legacy else 15) if at == 2 else 14) * 4.3) * 2.4e9
    kern = "k_sw2 pairs/wg %s" % os.environ.get("FSGPU_SW2_PAIRS", "default") if os.environ.get("FSGPU_SW_PROFILES", "0") not in ("", "0") else \
        "k_sw3 waves/wg %s" % os.environ.get("FSGPU_SW3_WAVES", "default")
    print("alignment-type %d %s: forward pass %.3f ms per batch of %d (%.4f ms/query), %.3e cells -> %.3f Tcell/s = %.3f of the issue bound; "
          "issued VALU fraction %.3f; records %d" % (at, kern, ms, NQ, ms / NQ, cells, cells / ms / 1e9, cells / (ms * 1e-3) / peak,
                                                     float(best[0][3]) * 4.3 / 1024 / 2.4e9 / (ms * 1e-3), sum(len(x) for x in r)), flush=True)
    s.close()
