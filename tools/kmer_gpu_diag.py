"""First-contact diagnostics of the k-mer GPU path vs the oracle: prints per-stage agreement instead of asserting."""
import sys, time, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import helpers as H, kmer_lib as K
from foldseek_amd import api, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
NQ = int(sys.argv[2]) if len(sys.argv) > 2 else 6
O = K.load_ora()
q3, qa = synth.make_queries(NQ, seed=1)
db = synth.make_db(N, (q3, qa), homologs_per_query=30, mask_frac=0.02)
targets = [db.seq(i, "3di", unmask=False) for i in range(db.n)]
ksub, pb = H.o_submat("MAT3DI", 8.0, -0.2); usub, _ = H.o_submat("MAT3DI", 2.0, -0.2)
t = time.time(); o = K.OraKpf(O, ksub, pb, usub, targets); print("oracle build %.1fs" % (time.time() - t), flush=True)
ctx = api.Context(0); ctx.load_db(db)
m8 = api.Matrix(0, 8.0, -0.2); m2 = api.Matrix(0, 2.0, -0.2)
t = time.time(); ctx.kmer_index_build(m8, kmer_thr=78); print("gpu index build %.2fs entries=%d" % (time.time() - t, ctx.kmer_index_entries), flush=True)
ooff, oseq, opos = o.index()
off, seq, pos, msk = ctx.kmer_index_reference_order(db.data3di.size)
print("entries", ctx.kmer_index_entries, int(ooff[-1]), "offsets equal:", bool((off == ooff).all()))
if len(seq) == len(oseq):
    print("entry seq equal:", bool((seq == oseq).all()), "pos equal:", bool((pos == opos).all()))
bad = [i for i in range(db.n) if not (msk[db.offsets[i]:db.offsets[i] + db.lengths[i]] == o.masked(i, int(db.lengths[i]))).all()]
print("masked mismatches:", len(bad), bad[:5])
badrows = []
for idx in [0, 1, 19, 20, 399, 400, 7999, 1234, 4321, 6789]:
    s, ix = ctx.kmer_row(idx); os_, oi = o.row(3, idx)
    if not ((s == os_).all() and (ix.astype(np.uint32) == oi).all()): badrows.append(idx)
print("bad 3-mer rows:", badrows, flush=True)
VAR = [dict(), dict(maxResListLen=50), dict(maxResListLen=50, bins=4), dict(maxResListLen=5), dict(maxResListLen=300, maxDbMatches=20000),
       dict(maxResListLen=100, maxDbMatches=9000, bins=8), dict(maxResListLen=300, maxDbMatches=5000, foundDiagonalsSize=40000),
       dict(compBias=0, maxResListLen=20, maxDbMatches=15000), dict(minDiagScoreThr=15, maxResListLen=2000)]
ident = np.full(NQ, -1, np.int64); ident[1] = 7
for kw in VAR:
    base = dict(maxResListLen=1000, bins=0, maxDbMatches=0, foundDiagonalsSize=0, compBias=1, minDiagScoreThr=30); base.update(kw)
    o.set(**base)
    orr, os_ = o.run(q3, ident)
    prep = [api.kmer_query_prepare(m8, m2, q, comp_bias=bool(base["compBias"]), scale=0.15, kmer_thr=78) for q in q3]
    t = time.time()
    res, status, stats = ctx.kmer_search(prep, identity=ident, max_res=base["maxResListLen"], min_diag=base["minDiagScoreThr"], bins=base["bins"],
                                         max_db_matches=base["maxDbMatches"], found_diagonals_size=base["foundDiagonalsSize"], l2_cache_size=2 << 20, want_stats=True)
    dt = time.time() - t
    print("==", kw, "wall %.1f ms  stages(ms)" % (dt * 1e3), ["%.3f" % x for x in ctx.kmer_stage_ms()])
    for q in range(NQ):
        a, b = res[q], orr[q]
        same = len(a) == len(b) and (a == b).all()
        print("  q%d L=%d status=%d n=%d/%d %s stats gpu=%s ora=%s" % (q, len(q3[q]), status[q], len(a), len(b), "OK" if same else "MISMATCH", stats[q].tolist(), os_[q].tolist()))
        if not same:
            n = min(len(a), len(b))
            d = np.nonzero(a[:n] != b[:n])[0][:4]
            print("     first diffs at", d.tolist(), "gpu", a[d].tolist(), "ora", b[d].tolist())
            sa, sb = set(a["id"].tolist()), set(b["id"].tolist())
            print("     ids only gpu", sorted(sa - sb)[:6], "only ora", sorted(sb - sa)[:6])
