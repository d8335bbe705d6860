import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from foldseek_amd import api, synth
db = synth.make_db(100000, None)
ctx = api.Context(0); ctx.load_db(db)
par = api.default_params(); par.alignmentType = 2
s = api.Search(ctx, par)
rng = np.random.default_rng(1)
ids = rng.choice(db.n, 1000, replace=False).astype(np.uint32)
for L in (150, 300, 400, 512, 513, 800, 1024, 1500, 3000):
    q3 = rng.integers(0, 20, L).astype(np.uint8); qa = rng.integers(0, 20, L).astype(np.uint8)
    s.align(qa, q3, ids)
    t = time.perf_counter()
    for _ in range(3):
        s.align(qa, q3, ids)
    dt = (time.perf_counter() - t) / 3
    st = s.stats()
    print("L=%4d align call %.3f ms  sw kernels %.3f ms  prepare %.3f gates %.3f" % (L, dt * 1e3, ctx.kernel_ms(1), st[2] * 1e3, st[4] * 1e3), flush=True)
for L in (150, 400, 513, 1024):
    qs = [rng.integers(0, 20, L).astype(np.uint8) for _ in range(3)]
    m8, m2 = api.Matrix(0, 8.0, -0.2), api.Matrix(0, 2.0, -0.2)
    if L == 150:
        ctx.kmer_index_build(m8, kmer_thr=78)
    prep = [api.kmer_query_prepare(m8, m2, q) for q in qs]
    ctx.kmer_search(prep)
    t = time.perf_counter(); ctx.kmer_search(prep); dt = time.perf_counter() - t
    print("kmer L=%4d: %.3f ms/query" % (L, dt / 3 * 1e3), flush=True)
