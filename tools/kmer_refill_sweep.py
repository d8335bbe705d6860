"""CPU-only sweep (needs oracle/_ref): the oracle restatement of the k-mer match-count mode (--diag-score 0) with databaseHits refills against the
compiled reference over random (max-seqs, bins, buffer sizes, bias, threshold) settings.  python tools/kmer_refill_sweep.py [seed] [rounds]
Round 3: seed 1, 80 rounds x 5 queries, 243 queries with refills, 13388 duplicated entries reproduced, 0 mismatches."""
import sys, numpy as np
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import test_kmer_oracle_vs_ref as T
g = T.world.__wrapped__(); w = next(g)
r, o, q3 = w["r"], w["o"], w["q3"]
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
def canon(a): return a[np.lexsort((a["diag"], a["id"], -a["score"]))]
bad = 0; nref = 0; ndup = 0; skipped = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 60):
    kw = dict(maxResListLen=int(rng.choice([5, 40, 300, 1000, 3000])), bins=int(rng.choice([0, 2, 4, 8, 16, 32])), maxDbMatches=int(rng.choice([1500, 2500, 4000, 7000, 12000, 30000])),
              foundDiagonalsSize=int(rng.choice([0, 0, 2500, 6000])), compBias=int(rng.integers(0, 2)), minDiagScoreThr=int(rng.choice([0, 1, 3, 10])), noDiagScore=1)
    r.set(**kw); o.set(**kw)
    ident = np.array([-1, 7, -1, 100, -1], np.int64)
    rr, rs, _ = r.run(q3, ident); orr, os_ = o.run(q3, ident)
    ok = True
    for q in range(T.NQ):
        if orr[q] is None: skipped += 1; continue
        same = len(rr[q]) == len(orr[q]) and (canon(rr[q]) == canon(orr[q])).all() and np.allclose(rs[q], os_[q])
        if not same: ok = False; print("MISMATCH", it, q, kw, len(rr[q]), len(orr[q]), rs[q], os_[q])
        nref += rs[q][2] > 0; ndup += len(rr[q]) - len(np.unique(rr[q]["id"]))
    bad += not ok
print("rounds bad", bad, "queries with refills", nref, "duplicated entries", ndup, "skipped(sort branch)", skipped)
