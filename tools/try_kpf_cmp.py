import sys, time, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from foldseek_amd import synth
import kmer_lib as K, helpers as H
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
NQ = int(sys.argv[2]) if len(sys.argv) > 2 else 6
kw = {}
for a in sys.argv[3:]:
    k, v = a.split("="); kw[k] = float(v) if "." in v else int(v)
q3, qa = synth.make_queries(NQ, seed=1)
db = synth.make_db(N, (q3, qa), homologs_per_query=30)
targets = [db.seq(i, "3di", unmask=False) for i in range(db.n)]
R = K.load_ref(); O = K.load_ora()
ksub, pb = H.o_submat("MAT3DI", 8.0, -0.2); usub, _ = H.o_submat("MAT3DI", 2.0, -0.2)
t = time.time(); r = K.RefKpf(R, targets, **kw); print("ref build", time.time() - t)
assert (r.submat(0).ravel() == ksub).all() and (r.submat(1).ravel() == usub).all()
t = time.time(); o = K.OraKpf(O, ksub, pb, usub, targets, **kw); print("ora build", time.time() - t)
ro, oo = r.offsets(), o.offsets()
print("offsets equal", (ro == oo).all(), ro[-1], oo[-1])
for i in range(0, db.n, max(1, db.n // 50)):
    assert (r.masked(i, len(targets[i])) == o.masked(i, len(targets[i]))).all(), i
rng = np.random.default_rng(5)
nz = np.nonzero(np.diff(ro.astype(np.int64)))[0]
for k in rng.choice(nz, 200):
    a, b = r.index_list(k), o.index_list(k)
    assert (a[0] == b[0]).all() and (a[1] == b[1]).all(), k
for idx in rng.integers(0, 8000, 20):
    a, b = r.row(3, idx), o.row(3, idx)
    assert (a[0] == b[0]).all() and (a[1] == b[1]).all(), idx
for _ in range(30):
    km = rng.integers(0, 20, 6).astype(np.uint8); thr = int(rng.integers(40, 120))
    a, b = r.kmer_list(km, thr), o.kmer_list(km, thr)
    assert len(a) == len(b) and (a == b).all(), (km, thr, len(a), len(b))
print("pieces ok")
ident = np.full(NQ, -1, np.int64); ident[0] = 5
t = time.time(); rr, rs, _ = r.run(q3, ident); print("ref run", time.time() - t)
t = time.time(); orr, os_ = o.run(q3, ident); print("ora run", time.time() - t)
for q in range(NQ):
    same = len(rr[q]) == len(orr[q]) and (rr[q] == orr[q]).all()
    print(q, len(q3[q]), len(rr[q]), len(orr[q]), "OK" if same else "MISMATCH", rs[q], os_[q])
    if not same:
        n = min(len(rr[q]), len(orr[q]))
        bad = np.nonzero(rr[q][:n] != orr[q][:n])[0][:5]
        print(" first diffs", bad, rr[q][bad], orr[q][bad])
