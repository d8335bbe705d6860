"""Very long queries against very long targets (the reference's limit is 65535 residues): gapless scores, hit lists, SW score / end
positions against the oracle, accepted alignment records of the batch path against the single-query path.  (From about 20 000 residues on the
reference's e-value network rates even a 72 000-bit self-like hit at e = 32, so nothing is accepted there -- the compiled reference does the same.)
usage: long_probe.py [L ...]   (65535 takes ~100 s per alignment type in the CPU oracle)"""
import sys, time, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from foldseek_amd import api, synth
import helpers as H
rng = np.random.default_rng(11)
bad = 0
for L in [int(x) for x in sys.argv[1:]] or [5000, 20000]:
    q3 = rng.integers(0, 20, L).astype(np.uint8); qa = rng.integers(0, 20, L).astype(np.uint8)
    h3, ha = synth._mutate(rng, q3, qa, 0.3, 0.02)
    h3, ha = h3[:65535], ha[:65535]
    seqs3 = [h3, rng.integers(0, 20, 300).astype(np.uint8), rng.integers(0, 20, min(65535, L + 777)).astype(np.uint8), q3[: L // 2].copy()]
    seqsa = [ha, rng.integers(0, 20, 300).astype(np.uint8), rng.integers(0, 20, len(seqs3[2])).astype(np.uint8), qa[: L // 2].copy()]
    lens = np.array([len(x) for x in seqs3], np.int32); order = np.argsort(lens, kind="stable"); lens = lens[order]
    off = np.zeros(len(lens) + 1, np.int64); off[1:] = np.cumsum((lens + 3) // 4 * 4)
    d3 = np.full(int(off[-1]), 20, np.uint8); da = np.full(int(off[-1]), 20, np.uint8)
    for new, old in enumerate(order):
        d3[off[new]:off[new] + lens[new]] = seqs3[old]; da[off[new]:off[new] + lens[new]] = seqsa[old]
    db = synth.PaddedDB(d3, da, off, lens)
    ctx = api.Context(0); ctx.load_db(db)
    for atype in (0, 2):
        par = api.default_params(); par.alignmentType = atype; par.addBacktrace = 1
        s = api.Search(ctx, par)
        t0 = time.time()
        hits = s.prefilter(q3)
        got = ctx.gapless_scores().astype(np.int32)
        want = H.o_ungapped_scores(q3, db, True)
        ok = (got == want).all()
        ids = np.arange(db.n, dtype=np.uint32)
        res, bt = s.align(qa, q3, ids, with_backtrace=True)
        f, r = s.last_sw(len(ids))
        pA, p3 = H.o_align_profiles(qa, q3, atype)[:2]
        for k in range(db.n):
            ta, t3 = H.target_seqs(db, k)
            w = H.o_sw(pA, p3, L, ta, t3)
            if (int(f[k]["score"]), int(f[k]["qEnd"]), int(f[k]["dbEnd"])) != (int(w["score"]), int(w["qEnd"]), int(w["dbEnd"])):
                ok = False; print("  SW MISMATCH", L, atype, k, f[k], w)
        resb, btb = s.align_batch([qa, qa[:400]], [q3, q3[:400]], [ids, ids], with_backtrace=True)
        same = len(resb[0]) == len(res) and all((resb[0][fl] == res[fl]).all() for fl in ("dbKey", "score", "qStartPos", "qEndPos", "dbStartPos", "dbEndPos", "alnLength")) and btb[0] == bt
        ok = ok and same
        bad += not ok
        print("L=%d atype=%d gapless %s, %d hits, %d alignments (lengths %s), batch==single %s, %.1f s %s" % (L, atype, (got == want).all(), len(hits), len(res), res["alnLength"].tolist(), same, time.time() - t0, "ok" if ok else "BAD"), flush=True)
        s.close()
    ctx.close()
print("long probe done:", bad, "bad")
