"""Randomised GPU-vs-oracle comparison of the gapless prefilter and the structure SW (single and multi-query launches).
usage: hot_fuzz.py [rounds] [seed]"""
import sys, numpy as np
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H
from foldseek_amd import api, synth
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for rd in range(rounds):
    n = int(rng.integers(1, 700))
    nq = int(rng.integers(1, 6))
    lens = [int(rng.choice([1, 2, 7, 15, 16, 17, 31, 33, 63, 64, 65, 100, 128, 129, 255, 256, 257, 383, 384, 385, 511, 512, 513, 600, 1025, 1100, 1537, 2049])) if rng.random() < 0.5
            else int(rng.integers(1, 900)) for _ in range(nq)]
    if nq >= 2 and rng.random() < 0.4:          # two or three row-tiled queries of one register class: multi-query tile launches
        L0 = int(rng.choice([560, 600, 1000, 1100, 1537]))
        for i in range(min(nq, int(rng.integers(2, 4)))):
            lens[i] = L0 + int(rng.integers(0, 6))
    q3 = [rng.integers(0, 21 if rng.random() < 0.3 else 20, size=L).astype(np.uint8) for L in lens]
    qa = [rng.integers(0, 21 if rng.random() < 0.3 else 20, size=L).astype(np.uint8) for L in lens]
    big = [q for q in zip(q3, qa) if len(q[0]) >= 30]
    db = synth.make_db(n, ([b[0] for b in big], [b[1] for b in big]) if big and n > 60 else None, seed=int(rng.integers(1 << 30)),
                       homologs_per_query=int(rng.integers(1, 12)), mask_frac=float(rng.choice([0, 0.05])), mean_len=float(rng.choice([30, 150, 350])), lo=1, hi=1400, stay=float(rng.choice([0.0, 0.0, 0.5, 0.8])))
    atype = int(rng.choice([0, 2])); cb = bool(rng.integers(0, 2)); max_res = int(rng.choice([1, 5, 50, 1000]))
    ctx = api.Context(0); ctx.load_db(db)
    par = api.default_params(); par.alignmentType = atype; par.compBiasCorrection = int(cb); par.maxResListLen = max_res
    s = api.Search(ctx, par)
    ok = True
    hit_lists = []
    idents = [int(rng.integers(n)) if rng.random() < 0.3 else -1 for _ in range(nq)]
    multi = s.prefilter_batch(q3, identity=np.array(idents, np.int64))      # multi-query scan launches (round 2)
    for i in range(nq):
        ident = idents[i]
        hits = s.prefilter(q3[i], ident)
        if not (len(multi[i]) == len(hits) and (multi[i] == hits).all()):
            ok = False; print("  MULTI-SCAN MISMATCH round", rd, "q", i, "L", lens[i], "n", n)
        want = H.o_prefilter_select(H.o_ungapped_scores(q3[i], db, cb), 30, ident, max_res)
        if not (len(hits) == len(want) and (hits["id"] == want["key"]).all() and (hits["score"] == want["score"]).all()):
            ok = False; print("  PREFILTER MISMATCH round", rd, "q", i, "L", lens[i], "n", n, "cb", cb, "max_res", max_res, len(hits), len(want))
        ids = rng.choice(n, size=min(n, int(rng.integers(1, 40))), replace=False).astype(np.uint32)
        hit_lists.append(ids)
    single = [s.align(qa[i], q3[i], hit_lists[i], with_backtrace=True) for i in range(nq)]
    sw_single = []
    for i in range(nq):
        s.align(qa[i], q3[i], hit_lists[i]); sw_single.append(s.last_sw(len(hit_lists[i])))
        pAf, p3f, _, _ = H.o_align_profiles(qa[i], q3[i], atype, cb)
        pAr, p3r, _, _ = H.o_align_profiles(qa[i][::-1].copy(), q3[i][::-1].copy(), atype, cb)
        f, r = sw_single[-1]
        for k, t in enumerate(hit_lists[i][:12]):
            ta, t3 = H.target_seqs(db, int(t))
            if len(t3) == 0:
                continue
            w = H.o_sw(pAf, p3f, lens[i], ta, t3); w2 = H.o_sw(pAr, p3r, lens[i], ta, t3)
            if (f[k]["score"], f[k]["qEnd"], f[k]["dbEnd"]) != (w["score"], w["qEnd"], w["dbEnd"]) or (r[k]["score"], r[k]["qEnd"], r[k]["dbEnd"]) != (w2["score"], w2["qEnd"], w2["dbEnd"]):
                ok = False; print("  SW MISMATCH round", rd, "q", i, "L", lens[i], "t", int(t), "Lt", len(t3), "atype", atype, f[k], w, r[k], w2)
    if rng.random() < 0.5:
        api.set_host_workers(int(rng.integers(0, 6)))
    batch, bts = s.align_batch(qa, q3, hit_lists, with_backtrace=True)
    for i in range(nq):
        r1, b1 = single[i]
        if not (len(batch[i]) == len(r1) and all((batch[i][f] == r1[f]).all() for f in ("dbKey", "score", "eval", "qStartPos", "qEndPos", "dbStartPos", "dbEndPos", "alnLength")) and bts[i] == b1):
            ok = False; print("  BATCH MISMATCH round", rd, "q", i, "L", lens[i])
    bad += 0 if ok else 1
    print("round %d n=%d lens=%s atype=%d cb=%d max_res=%d %s" % (rd, n, lens, atype, cb, max_res, "ok" if ok else "BAD"), flush=True)
    s.close(); ctx.close()
print("fuzz done: %d bad of %d rounds" % (bad, rounds))
