"""Randomised GPU-vs-oracle comparison of the k-mer prefilter: random DB shapes, query lengths, thresholds and every
order-shaping parameter.  usage: kmer_fuzz.py [rounds] [seed]"""
import sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import helpers as H, kmer_lib as K
from foldseek_amd import api, synth
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
O = K.load_ora()
ksub, pb = H.o_submat("MAT3DI", 8.0, -0.2); usub, _ = H.o_submat("MAT3DI", 2.0, -0.2)
m8, m2 = api.Matrix(0, 8.0, -0.2), api.Matrix(0, 2.0, -0.2)
bad = refused = merged = zeros = 0
for rd in range(rounds):
    n = int(rng.integers(1, 2500))
    nq = int(rng.integers(1, 7))
    mean = float(rng.choice([40, 120, 350]))
    q3, qa = synth.make_queries(nq, seed=int(rng.integers(1 << 30)), mean_len=mean, lo=5, hi=int(mean * 4))
    db = synth.make_db(n, (q3, qa), seed=int(rng.integers(1 << 30)), homologs_per_query=int(rng.integers(0, min(40, n) + 1)) if n > 50 else 0,
                       mask_frac=float(rng.choice([0.0, 0.02, 0.3])), mean_len=float(rng.choice([60, 200, 350])), lo=1, hi=1500, stay=float(rng.choice([0.0, 0.0, 0.5, 0.8])))
    targets = [db.seq(i, "3di", unmask=False) for i in range(db.n)]
    thr = int(rng.choice([78, 78, 78, 60, 96, 110]))
    spaced = int(rng.integers(0, 2))
    kw = dict(kmerThr=thr, spaced=spaced, maxResListLen=int(rng.choice([1, 3, 30, 300, 1000, 5000])), bins=int(rng.choice([0, 2, 4, 16, 64])),
              maxDbMatches=int(rng.choice([0, 0, 500, 3000, 20000])), minDiagScoreThr=int(rng.choice([30, 30, 10, 60, 1])),
              compBias=int(rng.integers(0, 2)), maskLowerCase=int(rng.integers(0, 2)), maskNrepeats=int(rng.choice([6, 6, 0, 2])))
    if rng.random() < 0.25:          # --diag-score 0: k-mer match counts as scores (cut 0 allowed)
        kw["noDiagScore"] = 1
        kw["minDiagScoreThr"] = int(rng.choice([0, 0, 1, 3]))
    if "noDiagScore" not in kw and rng.random() < 0.25:
        kw["foundDiagonalsSize"] = int(rng.choice([100, 200, 400, 1000, 3000]))      # findDuplicates cut short: replayed since round 4 (diagonal-score mode)
    delta = 0
    if "noDiagScore" not in kw and rng.random() < 0.2:
        # --min-ungapped-score 0 with diagonal scores: the ungapped matrix lowered on both sides so that score-0 elements exist (tests/test_kmer_gpu.py)
        delta = int(rng.choice([-6, -8, -10]))
        kw["minDiagScoreThr"] = 0
    qs = list(q3)
    if n > 3:
        qs.append(targets[int(rng.integers(n))].copy())            # a database member as query (identity path)
    ident = np.full(len(qs), -1, np.int64)
    if n > 3 and rng.random() < 0.7:
        ident[-1] = int(rng.integers(n))
    if len(sys.argv) > 3 and rd != int(sys.argv[3]):
        continue
    try:
        o = K.OraKpf(O, ksub, pb, (usub + delta).astype(usub.dtype), targets, **kw)
        ctx = api.Context(0); ctx.load_db(db)
        ctx.kmer_index_build(m8, kmer_thr=thr, spaced=spaced, mask_lower_case=kw["maskLowerCase"], mask_n_repeats=kw["maskNrepeats"])
        prep = [api.kmer_query_prepare(m8, m2, q, comp_bias=bool(kw["compBias"]), kmer_thr=thr, spaced=spaced) for q in qs]
        if delta:
            prep = [(a, b, np.clip(c.astype(np.int32) + delta, -128, 127).astype(np.int8)) for (a, b, c) in prep]
        res, status, stats = ctx.kmer_search(prep, identity=ident, max_res=kw["maxResListLen"], min_diag=kw["minDiagScoreThr"], bins=kw["bins"],
                                             max_db_matches=kw["maxDbMatches"], found_diagonals_size=kw.get("foundDiagonalsSize", 0), l2_cache_size=2 << 20, want_stats=True,
                                             kmer_score_only=bool(kw.get("noDiagScore", 0)))
        ok = True
        for i, q in enumerate(qs):
            b, st = o.query(q, int(ident[i]))
            if kw.get("noDiagScore", 0) and st[2] > 0 and b is not None and status[i] == 0:
                # a refill in the count mode: mergeScoreDuplicates is replayed on the device (k_kmer_merge_heads); a target can come out twice with
                # equal (score, id), whose relative order is whatever the reference's std::sort leaves -- such ties compare in diagonal order
                canon = lambda a: a[np.lexsort((a["diag"], a["id"], -a["score"].astype(np.int64)))]
                same = len(res[i]) == len(b) and (canon(res[i]) == canon(b)).all() and np.allclose(stats[i], st)
                merged += 1
            elif b is None:
                same = status[i] == 1          # the oracle does not model the std::sort branch
            elif status[i] == -2 or (status[i] == -1 and kw.get("noDiagScore", 0)):
                # FSGPU_KMER_E_CHUNKS (more than 255 databaseHits refills: only with the artificially small maxDbMatches of this fuzzer), or
                # FSGPU_KMER_E_OUTPUT in the count mode (findDuplicates cut short there stays a status) -- refused with a status, never answered wrongly
                refused += 1
                same = True
            elif status[i] < 0:
                same = False
            elif delta:
                canon = lambda a: a[np.lexsort((a["diag"], a["id"], -a["score"].astype(np.int64)))]
                same = len(res[i]) == len(b) and (canon(res[i]) == canon(b)).all() and (len(q) == 0 or np.allclose(stats[i], st))
                zeros += int((b["score"] == 0).sum())
            else:
                same = len(res[i]) == len(b) and (res[i] == b).all() and (len(q) == 0 or np.allclose(stats[i], st))
            if not same:
                ok = False
                print("  MISMATCH round", rd, "query", i, "L", len(q), "ident", ident[i], "status", status[i], "gpu n", len(res[i]), "ora n", None if b is None else len(b), kw, "n", n)
                print("    gpu", res[i][:5].tolist(), "ora", None if b is None else b[:5].tolist(), "stats", stats[i].tolist(), None if b is None else st.tolist())
        bad += 0 if ok else 1
        print("round %d n=%d nq=%d %s overflowed=%s %s" % (rd, n, len(qs), "ok" if ok else "BAD", stats[:, 2].tolist(), {k: kw.get(k, 0) for k in ("kmerThr", "spaced", "maxResListLen", "bins", "maxDbMatches", "noDiagScore", "foundDiagonalsSize")}), "delta", delta, flush=True)
        o.close(); ctx.close()
    except Exception as e:
        bad += 1
        print("round", rd, "EXCEPTION", repr(e), kw, "n", n, flush=True)
print("fuzz done: %d bad of %d rounds (%d queries refused with a status, %d count-mode queries with refills replayed, %d score-0 hits under a cut of 0)" % (bad, rounds, refused, merged, zeros))
