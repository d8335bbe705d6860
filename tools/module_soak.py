"""End-to-end soak of the drop-in module executables at DB scale: on-disk MMseqs DBs -> prefilter (k-mer) /
ungappedprefilter -> structurealign; wall times, peak RSS, spot checks against the library calls."""
import os, sys, time, subprocess, resource, tempfile, numpy as np
sys.path.insert(0, "/root/repo")
from foldseek_amd import api, synth, dbio
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
NQ = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
BIN = "/root/repo/foldseek_amd/bin/fsgpu-modules"
tmp = tempfile.mkdtemp(prefix="fssoak_", dir=os.environ.get("TMPDIR", "/tmp"))
q3, qa = synth.make_queries(NQ, seed=77, lo=150, hi=600)
t = time.time(); db = synth.make_db(N, (q3[:64], qa[:64])); print("db gen %.1fs" % (time.time() - t), flush=True)
qkeys = np.arange(NQ, dtype=np.uint32) + 5
qdb, tdb = os.path.join(tmp, "q"), os.path.join(tmp, "t")
dbio.write_seq_db(qdb, qa, qkeys); dbio.write_seq_db(qdb + "_ss", q3, qkeys)
dbio.write_padded_db(tdb, db, "aa"); dbio.write_padded_db(tdb + "_ss", db, "3di")
def run(args):
    t = time.time()
    p = subprocess.run([BIN] + args, capture_output=True, text=True)
    dt = time.time() - t
    ru = resource.getrusage(resource.RUSAGE_CHILDREN)
    print("%-18s rc=%d %.2fs  (%.3f ms/query)  maxrss %.0f MB %s" % (args[0], p.returncode, dt, dt / NQ * 1e3, ru.ru_maxrss / 1024, p.stderr.strip()[:200]), flush=True)
    assert p.returncode == 0
    return dt
pk, pu, a1, a2 = [os.path.join(tmp, x) for x in ("pref_k", "pref_u", "aln_k", "aln_u")]
run(["prefilter", qdb + "_ss", tdb + "_ss", pk, "--threads", "3"])
run(["structurealign", qdb, tdb, pk, a1, "--alignment-type", "2", "-a", "--threads", "3"])
if os.environ.get("SOAK_ALIGN_BATCH_SWEEP"):
    for ab in (8, 16, 32, 64):
        for f in (a1 + "_ab", a1 + "_ab.index", a1 + "_ab.dbtype"):
            if os.path.exists(f): os.remove(f)
        print("--align-batch", ab, end="  ")
        run(["structurealign", qdb, tdb, pk, a1 + "_ab", "--alignment-type", "2", "-a", "--threads", "3", "--align-batch", str(ab)])
        assert dbio.read_db(a1 + "_ab") == dbio.read_db(a1)
run(["ungappedprefilter", qdb + "_ss", tdb + "_ss", pu, "--threads", "3"])
run(["structurealign", qdb, tdb, pu, a2, "--alignment-type", "2", "-a", "--threads", "3"])
af = os.path.join(tmp, "aln_fused")
run(["search", qdb, tdb, af, "--prefilter-mode", "0", "--alignment-type", "2", "-a", "--threads", "3"])
_, afd = dbio.read_db(af)
_, dk = dbio.read_db(pk); _, du = dbio.read_db(pu); _, ak = dbio.read_db(a1); _, au = dbio.read_db(a2)
print("entries", len(dk), len(du), len(ak), len(au), "bytes", sum(map(len, dk.values())), sum(map(len, du.values())), sum(map(len, ak.values())), sum(map(len, au.values())))
# spot check 5 queries of each chain against the library
ctx = api.Context(0); ctx.load_db(db)
par = api.default_params(); par.alignmentType = 2; par.addBacktrace = 1
s = api.Search(ctx, par)
m8, m2 = api.Matrix(0, 8.0, -0.2), api.Matrix(0, 2.0, -0.2)
ctx.kmer_index_build(m8, kmer_thr=api.kmer_threshold(9.5, 6))
bad = 0
for i in (0, 1, 63, 500, NQ - 1):
    res, st = ctx.kmer_search([api.kmer_query_prepare(m8, m2, q3[i])], max_res=1000)
    exp = "".join(api.format_prefilter_hit(int(h["id"]), int(h["score"]), int(np.int16(h["diag"]))) for h in res[0])
    bad += dk[int(qkeys[i])].decode() != exp
    r, b = s.align(qa[i], q3[i], res[0]["id"], with_backtrace=True)
    bad += ak[int(qkeys[i])].decode() != "".join(s.format_result(r[j:j + 1], b[j], True) for j in range(len(r)))
    hits = s.prefilter(q3[i])
    bad += du[int(qkeys[i])].decode() != "".join(api.format_prefilter_hit(int(h["id"]), int(h["score"]), 0) for h in hits)
    r, b = s.align(qa[i], q3[i], hits["id"], with_backtrace=True)
    bad += au[int(qkeys[i])].decode() != "".join(s.format_result(r[j:j + 1], b[j], True) for j in range(len(r)))
# resident DB: gpuserver + ungappedprefilter --gpu-server 1 (one query in flight by protocol)
import signal
name = "fsgpu_soak_%d" % os.getpid()
t = time.time()
srv = subprocess.Popen([BIN, "gpuserver", tdb + "_ss", "--shm-name", name], stderr=subprocess.PIPE, text=True)
while not (os.path.exists("/dev/shm/" + name) and os.path.getsize("/dev/shm/" + name) > 0):
    assert srv.poll() is None
    time.sleep(0.02)
print("gpuserver up in %.2fs" % (time.time() - t), flush=True)
ps = os.path.join(tmp, "pref_served")
run(["ungappedprefilter", qdb + "_ss", tdb + "_ss", ps, "--gpu-server", "1", "--shm-name", name])
srv.send_signal(signal.SIGTERM); srv.wait(timeout=60)
_, dsv = dbio.read_db(ps)
print("served ungappedprefilter == direct:", dsv == du)
bad += dsv != du
print("fused search == prefilter + structurealign:", afd == ak)
bad += afd != ak
print("spot-check mismatches:", bad)
subprocess.run(["rm", "-rf", tmp])
assert bad == 0
