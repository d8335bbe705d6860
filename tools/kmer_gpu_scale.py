"""GPU k-mer prefilter vs the compiled reference at DB scale: parity of complete hit lists + timings."""
import sys, time, os, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import kmer_lib as K
from foldseek_amd import api, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
NQ = int(sys.argv[2]) if len(sys.argv) > 2 else 32
THREADS = int(sys.argv[3]) if len(sys.argv) > 3 else (os.cpu_count() or 8)
STAY = float(os.environ.get("FS_STAY", "0"))
R = K.load_ref()
q3, qa = synth.make_queries(NQ, seed=1)
if STAY > 0:
    rs = np.random.default_rng(99)
    q3 = [synth.sticky(rs, q, STAY) for q in q3]
t = time.time(); db = synth.make_db(N, (q3, qa), stay=STAY); print("db %.1fs residues=%d" % (time.time() - t, db.residues), flush=True)
ctx = api.Context(0); ctx.load_db(db)
m8 = api.Matrix(0, 8.0, -0.2); m2 = api.Matrix(0, 2.0, -0.2)
t = time.time(); ctx.kmer_index_build(m8, kmer_thr=78); print("gpu index build %.2fs entries=%d" % (time.time() - t, ctx.kmer_index_entries), flush=True)
prep = [api.kmer_query_prepare(m8, m2, q) for q in q3]
l2 = int(R.ref_l2_cache_size()) if R is not None else 0
for rep in range(3):
    t = time.time()
    res, status, stats = ctx.kmer_search(prep, max_res=1000, l2_cache_size=l2, want_stats=True)
    dt = time.time() - t
    print("gpu search %d queries: %.2f ms (%.3f ms/query) stages %s" % (NQ, dt * 1e3, dt * 1e3 / NQ, ["%.2f" % x for x in ctx.kmer_stage_ms()]), flush=True)
print("status", status.tolist(), "overflowed", stats[:, 2].tolist())
print("hits/query", stats[:, 1].astype(np.int64).tolist())
if R is not None:
    targets = [db.seq(i, "3di", unmask=False) for i in range(db.n)]
    t = time.time(); r = K.RefKpf(R, targets, threads=THREADS); print("ref index build %.1fs (%d threads)" % (time.time() - t, THREADS), flush=True)
    rr, rs, secs = r.run(q3, None, threads=THREADS)
    print("ref run %d threads: %.3fs -> %.2f ms/query throughput" % (THREADS, secs, secs / NQ * 1e3))
    rr1, rs1, secs1 = r.run(q3[:4], None, threads=1)
    print("ref run 1 thread: %.1f ms/query" % (secs1 / 4 * 1e3))
    bad = 0
    for q in range(NQ):
        same = len(res[q]) == len(rr[q]) and (res[q] == rr[q]).all()
        if not same:
            bad += 1
            n = min(len(res[q]), len(rr[q])); d = np.nonzero(res[q][:n] != rr[q][:n])[0][:3]
            print("  q%d L=%d MISMATCH n=%d/%d status=%d first diffs %s gpu %s ref %s" % (q, len(q3[q]), len(res[q]), len(rr[q]), status[q], d.tolist(), res[q][d].tolist(), rr[q][d].tolist()))
    print("parity vs compiled reference: %d/%d queries identical; ref bins=%s l2=%d" % (NQ - bad, NQ, rs[0, 3], l2))
