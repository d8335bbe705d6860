"""solo gapless kernel time vs query length for several workgroups-per-CU settings (FSGPU_GAPLESS_BLOCKS_PER_CU is read
at context creation)"""
import os, sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from foldseek_amd import api, synth
db = synth.make_db(100000, None)
rng = np.random.default_rng(1)
qs = {L: rng.integers(0, 20, L).astype(np.uint8) for L in (100, 150, 200, 250, 300, 350, 400, 450)}
for per_cu in (3, 4, 5, 6, 8):
    os.environ["FSGPU_GAPLESS_BLOCKS_PER_CU"] = str(per_cu)
    ctx = api.Context(0); ctx.load_db(db)
    s = api.Search(ctx, api.default_params())
    out = []
    for L, q in qs.items():
        s.prefilter(q)
        k = []
        for _ in range(6):
            s.prefilter(q); k.append(ctx.kernel_ms(0))
        out.append("L=%d %.3f" % (L, min(k)))
    print("WG/CU %d: " % per_cu + "  ".join(out), flush=True)
    s.close(); ctx.close()
