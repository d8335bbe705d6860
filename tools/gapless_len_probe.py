import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from foldseek_amd import api, synth
db = synth.make_db(100000, None)
ctx = api.Context(0); ctx.load_db(db)
s = api.Search(ctx, api.default_params())
rng = np.random.default_rng(1)
for L in (150, 300, 400, 500, 512, 513, 600, 800, 1024, 1500):
    q = rng.integers(0, 20, L).astype(np.uint8)
    s.prefilter(q)
    t = time.perf_counter()
    for _ in range(5):
        s.prefilter(q)
    dt = (time.perf_counter() - t) / 5
    print("L=%4d prefilter call %.3f ms  kernel %.3f ms" % (L, dt * 1e3, ctx.kernel_ms(0)), flush=True)
