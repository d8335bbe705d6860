import sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from foldseek_amd import api, synth
import helpers as H, kmer_lib as K
rng = np.random.default_rng(3)
def mk(lens):
    n = len(lens); lens = np.array(lens, np.int32)
    order = np.argsort(lens, kind="stable"); lens = lens[order]
    pad = (lens + 3) // 4 * 4; off = np.zeros(n + 1, np.int64); off[1:] = np.cumsum(pad)
    d3 = np.full(int(off[-1]), 20, np.uint8); da = np.full(int(off[-1]), 20, np.uint8)
    for i in range(n):
        d3[off[i]:off[i] + lens[i]] = rng.integers(0, 20, lens[i]); da[off[i]:off[i] + lens[i]] = rng.integers(0, 20, lens[i])
    return synth.PaddedDB(d3, da, off, lens)
m8, m2 = api.Matrix(0, 8.0, -0.2), api.Matrix(0, 2.0, -0.2)
q = rng.integers(0, 20, 120).astype(np.uint8); qa = rng.integers(0, 20, 120).astype(np.uint8)
for name, lens in (("n=1", [50]), ("n=2 tiny", [1, 2]), ("one long 32767", [32767, 100]), ("one 40000", [40000, 64]), ("mixed tiny", [1, 1, 3, 9, 10, 11, 300])):
    try:
        db = mk(lens)
        ctx = api.Context(0); ctx.load_db(db)
        par = api.default_params(); par.alignmentType = 2
        s = api.Search(ctx, par)
        hits = s.prefilter(q)
        want = H.o_prefilter_select(H.o_ungapped_scores(q, db, True), 30, -1, 1000)
        okp = len(hits) == len(want) and (hits["id"] == want["key"]).all() and (hits["score"] == want["score"]).all()
        ids = np.arange(db.n, dtype=np.uint32)
        res = s.align(qa, q, ids)
        resb = s.align_batch([qa, qa], [q, q], [ids, ids[:1]])
        msg = "prefilter ok=%s hits=%d aligned=%d batch=%d,%d" % (okp, len(hits), len(res), len(resb[0]), len(resb[1]))
        try:
            ctx.kmer_index_build(m8, kmer_thr=78)
            r, st = ctx.kmer_search([api.kmer_query_prepare(m8, m2, q)], max_res=10)
            o = K.OraKpf(K.load_ora(), *H.o_submat("MAT3DI", 8.0, -0.2), H.o_submat("MAT3DI", 2.0, -0.2)[0], [db.seq(i, "3di", unmask=False) for i in range(db.n)], maxResListLen=10)
            b, _ = o.query(q, -1); o.close()
            msg += " | kmer status=%s n=%d oracle=%d same=%s" % (st.tolist(), len(r[0]), len(b), len(r[0]) == len(b) and (r[0] == b).all())
        except api.FsgpuError as e:
            msg += " | kmer: " + str(e)[:90]
        print(name, "->", msg, flush=True)
        s.close(); ctx.close()
    except Exception as e:
        print(name, "EXCEPTION", repr(e)[:200], flush=True)
# empty DB
try:
    db = synth.PaddedDB(np.zeros(0, np.uint8), np.zeros(0, np.uint8), np.zeros(1, np.int64), np.zeros(0, np.int32))
    ctx = api.Context(0); ctx.load_db(db); print("n=0 load ok, size", ctx.n)
    try:
        s = api.Search(ctx, api.default_params()); print("n=0 prefilter", len(s.prefilter(q)))
    except Exception as e:
        print("n=0 search:", repr(e)[:120])
except Exception as e:
    print("n=0 EXCEPTION", repr(e)[:200])
