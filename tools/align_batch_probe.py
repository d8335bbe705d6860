import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from foldseek_amd import api, synth
q3, qa = synth.make_queries(64, seed=5000, lo=250, hi=450)
db = synth.make_db(100000, synth.make_queries(8, seed=1000, lo=250, hi=450))
ctx = api.Context(0); ctx.load_db(db)
par = api.default_params(); par.alignmentType = 0
s = api.Search(ctx, par)
rng = np.random.default_rng(1)
hits = [rng.choice(db.n, 1000, replace=False).astype(np.uint32) for _ in range(64)]
for rep in range(3):
    t = time.perf_counter(); r = s.align_batch(qa[:32], q3[:32], hits[:32]); dt = time.perf_counter() - t
    st = s.stats()
    print("batch32: %.2f ms total; prepare %.2f device %.2f gates %.2f backtrace %.2f ; sw kernels %.2f ms" % (dt * 1e3, st[2] * 1e3, st[3] * 1e3, st[4] * 1e3, st[5] * 1e3, ctx.kernel_ms(1)))
for rep in range(2):
    t = time.perf_counter()
    for i in range(32):
        s.align(qa[i], q3[i], hits[i])
    print("32 single aligns: %.2f ms" % ((time.perf_counter() - t) * 1e3))
