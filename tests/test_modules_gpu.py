"""GPU test (-m gpu) of the drop-in modules: ASCII / padded MMseqs databases on disk -> fsgpu-modules ungappedprefilter ->
structurealign -> result databases, compared with the library pipeline that test_golden.py pins to the reference."""
import os
import subprocess
import numpy as np
import pytest

from foldseek_amd import api, synth, dbio
import helpers

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "foldseek_amd", "bin", "fsgpu-modules")


def _expected(db, q3, qa, keys_t, atype, max_seqs):
    ctx = api.Context(0)
    ctx.load_db(db)
    par = api.default_params()
    par.alignmentType = atype
    par.maxResListLen = max_seqs
    par.addBacktrace = 1
    s = api.Search(ctx, par, keys=keys_t)
    pref, aln = [], []
    for i in range(len(q3)):
        hits = s.prefilter(q3[i])
        pref.append("".join(api.format_prefilter_hit(int(keys_t[h["id"]]), int(h["score"]), 0) for h in hits))
        res, bts = s.align(qa[i], q3[i], hits["id"], with_backtrace=True)
        aln.append("".join(s.format_result(res[k:k + 1], bts[k], True) for k in range(len(res))))
    s.close()
    ctx.close()
    return pref, aln


@pytest.mark.parametrize("padded", [False, True])
def test_modules_end_to_end(tmp_path, padded):
    rng = np.random.default_rng(31)
    q3, qa = synth.make_queries(5, seed=41, mean_len=150, lo=60, hi=300)
    db = synth.make_db(700, (q3, qa), seed=42, homologs_per_query=25, mean_len=180, lo=30, hi=600, mask_frac=0.02)
    qkeys = [100 + 7 * i for i in range(len(q3))]
    qdb = str(tmp_path / "query")
    dbio.write_seq_db(qdb, qa, qkeys)
    dbio.write_seq_db(qdb + "_ss", q3, qkeys)
    tdb = str(tmp_path / "target")
    if padded:
        dbio.write_padded_db(tdb, db, "aa")
        dbio.write_padded_db(tdb + "_ss", db, "3di")
        keys_t = np.arange(db.n, dtype=np.uint32)
    else:
        keys_t = (np.arange(db.n) * 3 + 5).astype(np.uint32)
        seqs3, masks, seqsa = [], [], []
        for i in range(db.n):
            raw = db.data3di[db.offsets[i]:db.offsets[i] + db.lengths[i]]
            seqs3.append(np.where(raw >= 32, raw - 32, raw).astype(np.uint8))
            masks.append(raw >= 32)
            seqsa.append(db.dataaa[db.offsets[i]:db.offsets[i] + db.lengths[i]])
        dbio.write_seq_db(tdb, seqsa, keys_t)
        dbio.write_seq_db(tdb + "_ss", seqs3, keys_t, masks)
    pref, aln = str(tmp_path / "pref"), str(tmp_path / "aln")
    for atype in (2, 0):
        for f in (pref, aln):
            for ext in ("", ".index", ".dbtype"):
                if os.path.exists(f + ext):
                    os.remove(f + ext)
        subprocess.check_call([BIN, "ungappedprefilter", qdb + "_ss", tdb + "_ss", pref, "--max-seqs", "150", "--threads", "2"])
        subprocess.check_call([BIN, "structurealign", qdb, tdb, pref, aln, "--alignment-type", str(atype), "-a", "--threads", "2", "-e", "10"])
        tp, dp = dbio.read_db(pref)
        ta, da = dbio.read_db(aln)
        assert tp & 0xffff == 7 and ta & 0xffff == 5
        epref, ealn = _expected(db, q3, qa, keys_t, atype, 150)
        assert sorted(dp.keys()) == sorted(qkeys) == sorted(da.keys())
        for i, k in enumerate(qkeys):
            assert dp[k].decode() == epref[i]
            assert da[k].decode() == ealn[i]
        assert sum(len(v) for v in da.values()) > 0


@pytest.mark.parametrize("padded", [False, True])
def test_prefilter_module_matches_oracle(tmp_path, padded):
    """fsgpu-modules prefilter (k-mer prefilter) over on-disk DBs == the oracle's QueryMatcher restatement, text for text;
    run once as a search (query DB != target DB) and once all-vs-all (same DB: identity hit first with score 65535)."""
    import kmer_lib as K
    q3, qa = synth.make_queries(4, seed=51, mean_len=200, lo=80, hi=400)
    db = synth.make_db(900, (q3, qa), seed=52, homologs_per_query=20, mean_len=200, lo=30, hi=600, mask_frac=0.02)
    targets = [db.seq(i, "3di", unmask=False) for i in range(db.n)]
    ksub, pb = helpers.o_submat("MAT3DI", 8.0, -0.2)
    usub, _ = helpers.o_submat("MAT3DI", 2.0, -0.2)
    l2 = os.sysconf("SC_LEVEL2_CACHE_SIZE") if "SC_LEVEL2_CACHE_SIZE" in os.sysconf_names else 0
    o = K.OraKpf(K.load_ora(), ksub, pb, usub, targets, maxResListLen=120, l2CacheSize=l2 if l2 > 0 else 262144)
    qkeys = [100 + 7 * i for i in range(len(q3))]
    qdb = str(tmp_path / "query_ss")
    dbio.write_seq_db(qdb, q3, qkeys)
    tdb = str(tmp_path / "target_ss")
    if padded:
        dbio.write_padded_db(tdb, db, "3di")
        keys_t = np.arange(db.n, dtype=np.uint32)
    else:
        keys_t = (np.arange(db.n) * 3 + 5).astype(np.uint32)
        seqs3 = [np.where(t >= 32, t - 32, t).astype(np.uint8) for t in targets]
        dbio.write_seq_db(tdb, seqs3, keys_t, [t >= 32 for t in targets])
    out = str(tmp_path / "pref")
    subprocess.check_call([BIN, "prefilter", qdb, tdb, out, "--max-seqs", "120", "-s", "9.5", "--threads", "2"])
    ty, d = dbio.read_db(out)
    assert ty & 0xffff == 7 and sorted(d.keys()) == sorted(qkeys)
    for i, k in enumerate(qkeys):
        hits, _ = o.query(q3[i], -1)
        exp = "".join(api.format_prefilter_hit(int(keys_t[h["id"]]), int(h["score"]), int(np.int16(h["diag"]))) for h in hits)
        assert d[k].decode() == exp, i
        assert len(hits) > 0
    # the workflow chain of structuresearch.sh with the k-mer prefilter: prefilter -> structurealign (3Di+AA) on disk
    qa_db = str(tmp_path / "query")
    dbio.write_seq_db(qa_db, qa, qkeys)
    os.rename(qdb, qa_db + "_ss"); os.rename(qdb + ".index", qa_db + "_ss.index"); os.rename(qdb + ".dbtype", qa_db + "_ss.dbtype")
    ta_db = str(tmp_path / "target")
    if padded:
        dbio.write_padded_db(ta_db, db, "aa")
    else:
        dbio.write_seq_db(ta_db, [db.dataaa[db.offsets[i]:db.offsets[i] + db.lengths[i]] for i in range(db.n)], keys_t)
    for ext in ("", ".index", ".dbtype"):
        os.rename(tdb + ext, ta_db + "_ss" + ext)
    tdb = ta_db + "_ss"
    aln = str(tmp_path / "aln")
    subprocess.check_call([BIN, "structurealign", qa_db, ta_db, out, aln, "--alignment-type", "2", "-a", "--threads", "2", "-e", "10"])
    ta_, da_ = dbio.read_db(aln)
    assert ta_ & 0xffff == 5
    ctx = api.Context(0)
    ctx.load_db(db)
    par = api.default_params()
    par.alignmentType = 2
    par.addBacktrace = 1
    s = api.Search(ctx, par, keys=keys_t)
    naln = 0
    for i, k in enumerate(qkeys):
        hits, _ = o.query(q3[i], -1)
        res, bts = s.align(qa[i], q3[i], hits["id"], with_backtrace=True)
        exp = "".join(s.format_result(res[j:j + 1], bts[j], True) for j in range(len(res)))
        assert da_[k].decode() == exp, i
        naln += len(res)
    assert naln > 0
    s.close()
    ctx.close()
    # fused module: prefilter + structurealign in one process must write the same two databases
    for mode, ref_pref in ((0, out), (1, None)):
        aln2, pref2 = str(tmp_path / f"aln_fused{mode}"), str(tmp_path / f"pref_fused{mode}")
        subprocess.check_call([BIN, "search", qa_db, ta_db, aln2, pref2, "--prefilter-mode", str(mode), "--max-seqs", "120", "--alignment-type", "2",
                               "-a", "--threads", "2", "-e", "10"])
        if ref_pref is None:
            ref_pref, ref_aln = str(tmp_path / "pref_u"), str(tmp_path / "aln_u")
            subprocess.check_call([BIN, "ungappedprefilter", qa_db + "_ss", ta_db + "_ss", ref_pref, "--max-seqs", "120", "--threads", "2"])
            subprocess.check_call([BIN, "structurealign", qa_db, ta_db, ref_pref, ref_aln, "--alignment-type", "2", "-a", "--threads", "2", "-e", "10"])
        else:
            ref_aln = aln
        _, p_a = dbio.read_db(pref2); _, p_b = dbio.read_db(ref_pref)
        _, a_a = dbio.read_db(aln2); _, a_b = dbio.read_db(ref_aln)
        assert p_a == p_b and a_a == a_b, mode
    if not padded:
        # all-vs-all on the first 200 targets' worth of queries is too slow for the oracle; check identity handling on 3
        out2 = str(tmp_path / "pref_self")
        subprocess.check_call([BIN, "prefilter", tdb, tdb, out2, "--max-seqs", "120", "--threads", "2"])
        ty2, d2 = dbio.read_db(out2)
        assert len(d2) == db.n
        for tid in (0, 411, 899):
            hits, _ = o.query(targets[tid], tid)
            exp = "".join(api.format_prefilter_hit(int(keys_t[h["id"]]), int(h["score"]), int(np.int16(h["diag"]))) for h in hits)
            assert d2[int(keys_t[tid])].decode() == exp
            assert exp.startswith("%d\t65535\t0\n" % keys_t[tid])
    o.close()


# ---- gpuserver: resident DB + the reference's shared-memory protocol ----
def _wait_for(path, proc, timeout=120.0):
    import time
    t0 = time.time()
    while not (os.path.exists(path) and os.path.getsize(path) > 0):
        assert proc.poll() is None, "gpuserver exited: " + (proc.stderr.read() if proc.stderr else "")
        assert time.time() - t0 < timeout, "gpuserver did not come up"
        time.sleep(0.05)


def test_gpuserver_roundtrip_and_raw_protocol(tmp_path):
    """(1) ungappedprefilter --gpu-server 1 through a resident gpuserver == the direct module, byte for byte;
    (2) a client that knows nothing but the reference's byte layout (GpuUtil.h) and state machine gets the same hits;
    (3) SIGTERM: the server flags serverExit and unlinks the block."""
    import mmap, signal, struct, time
    q3, qa = synth.make_queries(6, seed=77, mean_len=160, lo=40, hi=420)
    db = synth.make_db(900, (q3, qa), seed=78, homologs_per_query=30, mean_len=170, lo=30, hi=700, mask_frac=0.02)
    qkeys = [3 + 5 * i for i in range(len(q3))]
    qdb = str(tmp_path / "query_ss")
    dbio.write_seq_db(qdb, q3, qkeys)
    tdb = str(tmp_path / "target_ss_pad")
    dbio.write_padded_db(tdb, db, "3di")
    direct, served = str(tmp_path / "direct"), str(tmp_path / "served")
    subprocess.check_call([BIN, "ungappedprefilter", qdb, tdb, direct, "--max-seqs", "120"])
    name = "fsgpu_test_%d" % os.getpid()
    srv = subprocess.Popen([BIN, "gpuserver", tdb, "--max-seqs", "120", "--max-seq-len", "2000", "--shm-name", name], stderr=subprocess.PIPE, text=True)
    try:
        _wait_for("/dev/shm/" + name, srv)
        subprocess.check_call([BIN, "ungappedprefilter", qdb, tdb, served, "--max-seqs", "120", "--gpu-server", "1", "--shm-name", name])
        (t1, d1), (t2, d2) = dbio.read_db(direct), dbio.read_db(served)
        assert t1 == t2 and sorted(d1) == sorted(d2) == sorted(qkeys)
        for k in qkeys:
            assert d1[k] == d2[k]
        assert sum(len(v) for v in d1.values()) > 0
        # raw client: header = {u32 maxSeqLen, u32 maxResListLen, i32 state, u8 serverExit (+3 pad), u32 queryOffset, queryLen,
        # resultsOffset, resultLen, profileOffset}; states IDLE 0, RESERVED 1, READY 2, DONE 3
        fd = os.open("/dev/shm/" + name, os.O_RDWR)
        mm = mmap.mmap(fd, 0)
        os.close(fd)
        max_seq_len, max_res = struct.unpack_from("<II", mm, 0)
        assert (max_seq_len, max_res) == (2000, 120) and len(mm) == 36 + 2000 + 16 * 120 + 21 * 2000
        q_off, _, r_off, _, p_off = struct.unpack_from("<IIIII", mm, 16)
        assert (q_off, r_off, p_off) == (36, 36 + 2000, 36 + 2000 + 16 * 120)
        m = api.Matrix(0, 2.0)
        ctx = api.Context(0)
        ctx.load_db(db)
        for qi in (0, 3):
            L = len(q3[qi])
            pssm, cap = api.prefilter_profile(m, q3[qi], True, 0.15)
            assert struct.unpack_from("<i", mm, 8)[0] == 0
            struct.pack_into("<i", mm, 8, 1)                                   # RESERVED
            mm[q_off:q_off + L] = q3[qi].tobytes()
            mm[p_off:p_off + 21 * L] = np.ascontiguousarray(pssm, dtype=np.int8).tobytes()
            struct.pack_into("<I", mm, 20, L)
            struct.pack_into("<i", mm, 8, 2)                                   # READY
            t0 = time.time()
            while struct.unpack_from("<i", mm, 8)[0] != 3:
                assert time.time() - t0 < 60 and srv.poll() is None
            n = struct.unpack_from("<I", mm, 28)[0]
            res = np.frombuffer(mm[r_off:r_off + 16 * n], dtype=np.dtype([("id", "<u4"), ("score", "<i4"), ("qEnd", "<i4"), ("dbEnd", "<i4")]))
            struct.pack_into("<i", mm, 8, 0)                                   # IDLE
            want = helpers.o_ungapped_scores(q3[qi], db, True)
            sel = helpers.o_prefilter_select(want, -1, -1, 120)               # server side: no threshold, top max-seqs
            assert n == len(sel) == 120
            assert (res["id"] == sel["key"]).all() and (res["score"] == sel["score"]).all()
        ctx.close()
        srv.send_signal(signal.SIGTERM)
        srv.wait(timeout=60)
        assert mm[12] == 1                                                     # serverExit
        assert not os.path.exists("/dev/shm/" + name)
        mm.close()
    finally:
        if srv.poll() is None:
            srv.kill()
        if os.path.exists("/dev/shm/" + name):
            os.remove("/dev/shm/" + name)


def test_db_replication_and_gpus_option(tmp_path):
    """fsgpu_db_broadcast: a context that received the DB through the replication path (peer copies here -- one GPU on
    the box, so the second copy lives on the same device and RCCL is not involved) answers exactly like the source;
    `--gpus 1` is the single-device path of the modules, asking for more devices than visible fails loudly."""
    q3, qa = synth.make_queries(3, seed=91, mean_len=180, lo=60, hi=350)
    db = synth.make_db(800, (q3, qa), seed=92, homologs_per_query=20, mask_frac=0.02)
    src = api.Context(0)
    src.load_db(db)
    dst = api.Context(0)
    os.environ["FSGPU_NO_RCCL"] = "1"
    try:
        assert src.broadcast_db_to([dst]) is False
    finally:
        del os.environ["FSGPU_NO_RCCL"]
    assert dst.n == src.n and dst.residues == src.residues
    par = api.default_params()
    par.alignmentType = 2
    sa, sb = api.Search(src, par), api.Search(dst, par)
    m8, m2 = api.Matrix(0, 8.0, -0.2), api.Matrix(0, 2.0, -0.2)
    dst.kmer_index_build(m8, kmer_thr=api.kmer_threshold(9.5, 6))
    src.kmer_index_build(m8, kmer_thr=api.kmer_threshold(9.5, 6))
    for i in range(3):
        ha, hb = sa.prefilter(q3[i]), sb.prefilter(q3[i])
        assert (ha["id"] == hb["id"]).all() and (ha["score"] == hb["score"]).all() and len(ha) > 0
        ra, rb = sa.align(qa[i], q3[i], ha["id"]), sb.align(qa[i], q3[i], hb["id"])
        assert ra.tobytes() == rb.tobytes()
        ka, _ = src.kmer_search([api.kmer_query_prepare(m8, m2, q3[i])], max_res=200)
        kb, _ = dst.kmer_search([api.kmer_query_prepare(m8, m2, q3[i])], max_res=200)
        assert ka[0].tobytes() == kb[0].tobytes()
    sa.close(); sb.close(); dst.close(); src.close()
    # module level
    qdb, tdb = str(tmp_path / "q_ss"), str(tmp_path / "t_ss_pad")
    dbio.write_seq_db(qdb, q3, [1, 2, 3])
    dbio.write_padded_db(tdb, db, "3di")
    a, b = str(tmp_path / "a"), str(tmp_path / "b")
    subprocess.check_call([BIN, "ungappedprefilter", qdb, tdb, a, "--max-seqs", "100"])
    subprocess.check_call([BIN, "ungappedprefilter", qdb, tdb, b, "--max-seqs", "100", "--gpus", "1", "--threads", "2"])
    assert dbio.read_db(a) == dbio.read_db(b)
    ngpu = api.lib().fsgpu_device_count()
    r = subprocess.run([BIN, "ungappedprefilter", qdb, tdb, str(tmp_path / "c"), "--gpus", str(ngpu + 1)], capture_output=True, text=True)
    assert r.returncode != 0 and "--gpus" in r.stderr
    if ngpu > 1:       # more than one device on this box: the real thing, RCCL broadcast included
        subprocess.check_call([BIN, "ungappedprefilter", qdb, tdb, str(tmp_path / "d"), "--max-seqs", "100", "--gpus", "all"])
        assert dbio.read_db(a) == dbio.read_db(str(tmp_path / "d"))


def test_single_gpu_paths_run_through_rccl_when_required(tmp_path):
    """FSGPU_REQUIRE_RCCL=1 on a one-GPU box (what can be checked here of the 8-GPU replication path): the library's RCCL branch runs on the
    device through a one-rank communicator (fsgpu_rccl_selfcheck: librccl loads, ncclCommInitAll + a grouped ncclBroadcast on the context's
    stream); `fsgpu-modules search --gpus all` performs that check, prints the host budget it runs with, and fails when RCCL is switched off;
    `bench.py --gpus 1` puts the target DB through a one-rank RCCL process group and says so on its line."""
    import json
    import sys
    ctx = api.Context(0)
    ctx.rccl_selfcheck()
    ctx.close()
    q3, qa = synth.make_queries(3, seed=191, mean_len=180, lo=60, hi=350)
    db = synth.make_db(800, (q3, qa), seed=192, homologs_per_query=20, mask_frac=0.02)
    for name, which in (("t_ss", "3di"), ("t", "aa")):
        dbio.write_seq_db_from_padded(str(tmp_path / name), db, which)
    dbio.write_seq_db(str(tmp_path / "q_ss"), q3, [1, 2, 3])
    dbio.write_seq_db(str(tmp_path / "q"), qa, [1, 2, 3])
    cmd = [BIN, "search", "q", "t", "aln", "--gpus", "all", "--threads", "4", "-s", "9.5", "--max-seqs", "100", "--alignment-type", "2", "--sort-by-structure-bits", "0"]
    env = dict(os.environ, FSGPU_REQUIRE_RCCL="1")
    r = subprocess.run(cmd, cwd=tmp_path, env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    ngpu = api.lib().fsgpu_device_count()
    assert ("RCCL self-check" in r.stderr and "usedRccl 1" in r.stderr) if ngpu == 1 else "RCCL broadcast" in r.stderr, r.stderr[-2000:]
    assert "host budget:" in r.stderr and "feeder thread(s) per GPU" in r.stderr
    plain = subprocess.run(cmd[:4] + ["aln2"] + cmd[5:], cwd=tmp_path, capture_output=True, text=True)
    assert plain.returncode == 0 and dbio.read_db(str(tmp_path / "aln")) == dbio.read_db(str(tmp_path / "aln2"))
    if ngpu == 1:
        bad = subprocess.run(cmd, cwd=tmp_path, env=dict(env, FSGPU_NO_RCCL="1"), capture_output=True, text=True)
        assert bad.returncode == 0            # FSGPU_NO_RCCL is the explicit opt-out: no check, no failure
    b = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--targets", "20000", "--no-kmer", "--type2-steps", "0",
                        "--allvsall-steps", "0", "--fullrange-steps", "0", "--no-cpu-baseline"], env=env, capture_output=True, text=True, cwd=ROOT)
    assert b.returncode == 0, b.stderr[-3000:]
    line = json.loads([l for l in b.stdout.splitlines() if l.startswith("{")][-1])
    assert line["broadcast_backend"] == "nccl" and line["rccl_ranks"] == 1 and line["n_gpus"] == 1 and line["value"] > 0
