"""CPU tests of the module layer that need no GPU: makepaddedseqdb against the padded layout, usage errors."""
import ctypes as C
import os
import numpy as np
import pytest
from foldseek_amd import api, synth, dbio


def _call(fn, args):
    arr = (C.c_char_p * len(args))(*[a.encode() for a in args])
    f = getattr(C.CDLL(api.LIB_PATH), fn)
    f.restype = C.c_int
    return f(len(args), arr)


def test_makepaddedseqdb_layout(tmp_path):
    rng = np.random.default_rng(8)
    lens = [5, 33, 8, 8, 100, 1, 64, 33]
    seqs = [rng.integers(0, 21, size=l).astype(np.uint8) for l in lens]
    masks = [rng.random(l) < 0.2 for l in lens]
    keys = [10, 11, 12, 13, 14, 15, 16, 17]
    src = str(tmp_path / "db_ss")
    dbio.write_seq_db(src, seqs, keys, masks)
    dst = str(tmp_path / "db_ss_pad")
    assert _call("fsmod_makepaddedseqdb", [src, dst]) == 0
    t, data, offsets, lengths = dbio.read_padded_db(dst)
    assert t & 0xffff == 0 and (t >> 16) & 8
    # ascending length; ties: the reference iterates a (length desc, id asc) order backwards -> id descending
    order = sorted(range(len(lens)), key=lambda i: (lens[i], -i))
    assert list(lengths) == [lens[i] for i in order]
    assert (offsets % 4 == 0).all()
    for new, old in enumerate(order):
        got = data[offsets[new]:offsets[new] + lengths[new]]
        want = seqs[old] + 32 * masks[old].astype(np.uint8)
        assert (got == want).all()
        pad = data[offsets[new] + lengths[new]:offsets[new + 1]]
        assert (pad == 20).all()
    lookup = [l.split() for l in open(dst + ".lookup")]
    assert [int(x[1]) for x in lookup] == [keys[i] for i in order]


def test_module_usage_errors(tmp_path):
    assert _call("fsmod_ungappedprefilter", ["only-one-arg"]) != 0
    assert _call("fsmod_structurealign", ["a", "b"]) != 0
    assert _call("fsmod_makepaddedseqdb", [str(tmp_path / "missing"), str(tmp_path / "out")]) != 0


def test_prefilter_module_argument_errors(tmp_path):
    """option handling of the k-mer prefilter module that is decided before any device call"""
    assert _call("fsmod_prefilter", ["a", "b"]) != 0
    rng = np.random.default_rng(3)
    seqs = [rng.integers(0, 20, size=40).astype(np.uint8) for _ in range(4)]
    src = str(tmp_path / "db_ss")
    dbio.write_seq_db(src, seqs, [1, 2, 3, 4])
    out = str(tmp_path / "out")
    assert _call("fsmod_prefilter", [src, src, out, "-k", "7"]) != 0                    # only k = 6 on the device path
    assert _call("fsmod_prefilter", [src, src, out, "--mask", "1"]) != 0                  # tantan masking is not on the device path
    assert _call("fsmod_prefilter", [src, src, out, "--exact-kmer-matching", "1"]) != 0
    assert _call("fsmod_prefilter", [str(tmp_path / "missing"), src, out]) != 0
    assert not os.path.exists(out + ".index")


def test_kmer_threshold_and_query_prepare():
    """Prefiltering::getKmerThreshold values and the per-position thresholds / ungapped profile against the oracle's
    arithmetic (bias rounding rules of QueryMatcher.cpp:267-268 and UngappedAlignment.cpp:399-403)."""
    import helpers
    assert api.kmer_threshold(9.5, 6) == 78 and api.kmer_threshold(7.5, 6) == 96 and api.kmer_threshold(9.5, 7) == 90
    m8, m2 = api.Matrix(0, 8.0, -0.2), api.Matrix(0, 2.0, -0.2)
    ksub, pb = helpers.o_submat("MAT3DI", 8.0, -0.2)
    usub, _ = helpers.o_submat("MAT3DI", 2.0, -0.2)
    rng = np.random.default_rng(12)
    q = rng.integers(0, 21, size=133).astype(np.uint8)
    seq, thr, prof = api.kmer_query_prepare(m8, m2, q, comp_bias=True, scale=0.15, kmer_thr=78)
    cbf, _ = helpers.o_round_bias(ksub, pb, q, 0.15)
    pos = [0, 1, 3, 5, 8, 9]
    assert len(thr) == len(q) - 9
    for i in range(len(thr)):
        bc = np.float32(0)
        for z in pos:
            bc = np.float32(bc + cbf[i + z])
        b = int(np.trunc(np.float64(bc) - 0.5)) if bc < 0 else int(np.trunc(np.float64(bc) + 0.5))
        assert thr[i] == max(78 - b, 0)
    us = usub.reshape(21, 21)
    for p in range(len(q)):
        c = np.float32(cbf[p])
        c4 = np.float32(np.float64(np.float32(c / np.float32(4))) + (-0.5 if c < 0 else 0.5))
        assert (prof[p] == (us[q[p]] + int(np.trunc(c4))).astype(np.int8)).all()
    _, thr0, prof0 = api.kmer_query_prepare(m8, m2, q, comp_bias=False)
    assert (thr0 == 78).all() and (prof0 == us[q].astype(np.int8)).all()
    assert len(api.kmer_query_prepare(m8, m2, q[:7])[1]) == 0


# ---- gpuserver protocol: constants against the reference's own header, failure modes without a GPU ----
def _lib():
    lib = C.CDLL(api.LIB_PATH)
    lib.fshost_gpu_shm_bytes.restype = C.c_size_t
    lib.fshost_gpu_shm_bytes.argtypes = [C.c_uint, C.c_uint]
    lib.fshost_gpu_shm_name.restype = C.c_int
    lib.fshost_gpu_shm_name.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t]
    return lib


def test_gpu_shm_layout_and_size():
    """GPUSharedMemory (M/src/commons/GpuUtil.h:9-39) + Marv::Result: header offsets, block size"""
    lib = _lib()
    out = (C.c_uint * 11)()
    lib.fshost_gpu_shm_layout(out)
    assert list(out) == [36, 0, 4, 8, 12, 16, 20, 24, 28, 32, 16]
    for (msl, mrl) in ((65535, 1000), (1, 1), (32000, 300)):
        assert lib.fshost_gpu_shm_bytes(msl, mrl) == 36 + msl + 16 * mrl + 21 * msl
    import oracle_lib
    ref = oracle_lib.load_ref()
    if ref is not None and hasattr(ref, "ref_gpu_shm_layout"):
        want = (C.c_uint * 11)()
        ref.ref_gpu_shm_layout(want)
        assert list(out) == list(want)
        ref.ref_gpu_shm_bytes.restype = C.c_size_t
        ref.ref_gpu_shm_bytes.argtypes = [C.c_uint, C.c_uint]
        assert ref.ref_gpu_shm_bytes(65535, 1000) == lib.fshost_gpu_shm_bytes(65535, 1000)
        st = (C.c_int * 4)()
        ref.ref_gpu_shm_states(st)
        assert list(st) == [0, 1, 2, 3]             # IDLE, RESERVED, READY, DONE: the values the client/server code uses


def test_gpu_shm_name(tmp_path):
    """name = decimal Util::hash (h = 31 h + c over plain chars) of realpath(db minus .idx) + visible devices + version"""
    lib = _lib()
    db = tmp_path / "tdb_ss_pad"
    db.write_bytes(b"x")
    link = tmp_path / "link_ss_pad"
    os.symlink(db, link)

    buf = C.create_string_buffer(64)
    for path, dev, ver in ((str(db), None, "abc123"), (str(link), b"0,1", "v\xc3\xa9rsion"), (str(db) + ".idx", b"3", "")):
        n = lib.fshost_gpu_shm_name(path.encode(), dev, ver.encode("latin-1") if isinstance(ver, str) else ver, buf, 64)
        assert n > 0
        real = os.path.realpath(str(db))
        s = real.encode() + (dev or b"") + ver.encode("latin-1")
        h = 0
        for ch in s:
            c = ch - 256 if ch >= 128 else ch          # plain (signed) char, like the reference
            h = (h * 31 + c) & 0xFFFFFFFFFFFFFFFF
        assert buf.value.decode() == str(h)
        import oracle_lib
        ref = oracle_lib.load_ref()
        if ref is not None and hasattr(ref, "ref_util_hash"):
            ref.ref_util_hash.restype = C.c_size_t
            ref.ref_util_hash.argtypes = [C.c_char_p, C.c_size_t]
            assert str(ref.ref_util_hash(s, len(s))) == buf.value.decode()
    assert lib.fshost_gpu_shm_name(str(db).encode(), None, b"v", buf, 2) < 0     # buffer too small


def test_gpuserver_and_client_fail_loudly(tmp_path):
    """no GPU here: the server must refuse to start (no CPU fallback); a client without a server must say so"""
    import subprocess
    rng = np.random.default_rng(5)
    seqs = [rng.integers(0, 20, size=50).astype(np.uint8) for _ in range(6)]
    src = str(tmp_path / "db_ss")
    dbio.write_seq_db(src, seqs, list(range(6)))
    exe = os.path.join(os.path.dirname(api.LIB_PATH), "bin", "fsgpu-modules")
    assert _call("fsmod_gpuserver", []) != 0
    assert _call("fsmod_gpuserver", [str(tmp_path / "missing")]) != 0
    name = "fsgpu_test_nogpu_%d" % os.getpid()
    srv = subprocess.Popen([exe, "gpuserver", src, "--shm-name", name], stderr=subprocess.PIPE, text=True)
    try:
        srv.wait(timeout=20)
        assert srv.returncode != 0 and "GPU" in srv.stderr.read()          # no device: refuses, never serves from the CPU
        assert not os.path.exists("/dev/shm/" + name)
    except subprocess.TimeoutExpired:                                       # a GPU is present (test run on a GPU box): it serves
        import signal
        srv.send_signal(signal.SIGTERM)
        srv.wait(timeout=60)
        assert not os.path.exists("/dev/shm/" + name)
    r = subprocess.run([exe, "ungappedprefilter", src, src, str(tmp_path / "out"), "--gpu-server", "1", "--gpu-server-wait-timeout", "0",
                        "--shm-name", "fsgpu_test_absent_%d" % os.getpid()], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "gpuserver" in r.stderr and "not found" in r.stderr


def test_indexdb_needs_the_device_only_for_the_kmer_table(tmp_path):
    """`indexdb --index-subset 2` (sequence / header DBs + masked lookup) is file assembly and runs anywhere; with the k-mer table
    (`--index-subset 5`) the table comes from the device build -- without a GPU the module stops with the device error and writes no
    index (no CPU fallback); argument errors are reported before anything is opened"""
    import subprocess
    import torch
    rng = np.random.default_rng(8)
    seqs = [rng.integers(0, 20, size=int(L)).astype(np.uint8) for L in rng.integers(20, 90, size=12)]
    keys = list(range(3, 3 + 12))
    for name in ("t", "t_ss"):
        dbio.write_seq_db(str(tmp_path / name), seqs, keys)
        with open(tmp_path / f"{name}_h", "wb") as f, open(tmp_path / f"{name}_h.index", "w") as fi:
            off = 0
            for k in keys:
                b = f"h{k}".encode() + b"\n\0"
                f.write(b); fi.write(f"{k}\t{off}\t{len(b)}\n"); off += len(b)
        np.array([12], np.int32).tofile(str(tmp_path / f"{name}_h.dbtype"))
    exe = os.path.join(os.path.dirname(api.LIB_PATH), "bin", "fsgpu-modules")
    r = subprocess.run([exe, "indexdb", "t", "t", "--index-subset", "2", "--mask-lower-case", "1", "--mask-n-repeat", "6"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    idx = {int(l.split()[0]): (int(l.split()[1]), int(l.split()[2])) for l in open(tmp_path / "t.idx.index")}
    assert set(idx) == {0, 1, 2, 5, 6, 7, 8, 13, 14, 15, 16, 18, 19, 20, 21, 22, 23} and all(o % 4096 == 0 for o, _ in idx.values())
    data = open(tmp_path / "t.idx", "rb").read()
    assert data[idx[0][0]:idx[0][0] + 4] == b"fs1\0" and np.frombuffer(data, np.int32, 12, idx[1][0]).tolist()[:4] == [65535, 0, 1, 21]
    assert np.frombuffer(data, np.uint64, 1, idx[13][0])[0] == 12 and int(np.frombuffer(data, np.int64, 1, idx[15][0])[0]) == sum(len(s) for s in seqs)
    r = subprocess.run([exe, "indexdb", "t", "other"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode != 0 and "not implemented" in r.stderr
    r = subprocess.run([exe, "indexdb", "t_ss", "t_ss", "--index-subset", "5", "--index-dbsuffix", "_ss"], cwd=tmp_path, capture_output=True, text=True)
    if not torch.cuda.is_available():
        assert r.returncode != 0 and "GPU" in r.stderr and not os.path.exists(tmp_path / "t_ss.idx.index") and not os.path.exists(tmp_path / "t_ss.idx")
    else:
        assert r.returncode == 0 and os.path.exists(tmp_path / "t_ss.idx.index")


def _golden_copy(w):
    import shutil
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scop_v1")
    for f in ("db", "db.index", "db.dbtype", "db_h", "db_h.index", "db_h.dbtype", "db_ss", "db_ss.index", "db_ss.dbtype", "db.lookup", "db.source"):
        shutil.copy(os.path.join(gold, f), os.path.join(w, f))
    return sorted(int(l.split()[0]) for l in open(os.path.join(w, "db.index")))


ALL_RECORD_COLUMNS = ("query,target,evalue,gapopen,pident,fident,nident,qstart,qend,qlen,tstart,tend,tlen,alnlen,bits,cigar,qseq,tseq,qheader,theader,"
                      "qaln,taln,mismatch,qcov,tcov,qset,qsetid,tset,tsetid,q3di,t3di,q3dialn,t3dialn,prob,empty")


@pytest.mark.parametrize("tag,rec,ok,msg", [
    ("intact", "{t}\t100\t0.5\t1e-5\t0\t9\t{ql}\t0\t9\t{tl}\t10M\n", True, ""),
    ("no trailing newline", "{t}\t100\t0.5\t1e-5\t0\t9\t{ql}\t0\t9\t{tl}\t10M", True, ""),
    ("gaps only", "{t}\t100\t0.5\t1e-5\t0\t9\t{ql}\t0\t9\t{tl}\t10I10D\n", True, ""),
    ("backtrace longer than both entries", "{t}\t100\t0.5\t1e-5\t0\t9\t{ql}\t0\t9\t{tl}\t5000M\n", False, "backtrace longer"),
    ("run length that overflows", "{t}\t100\t0.5\t1e-5\t0\t9\t{ql}\t0\t9\t{tl}\t99999999999999999999M\n", False, "backtrace longer"),
    ("start outside the query", "{t}\t100\t0.5\t1e-5\t100000\t200000\t{ql}\t0\t9\t{tl}\t10M\n", False, "leaves the query"),
    ("start outside the target", "{t}\t100\t0.5\t1e-5\t0\t9\t{ql}\t70000\t90000\t{tl}\t10M\n", False, "leaves the target"),
    ("backtrace runs off the end", "{t}\t100\t0.5\t1e-5\t{qlm5}\t9\t{ql}\t0\t9\t{tl}\t10M\n", False, "leaves the query"),
    ("negative start", "{t}\t100\t0.5\t1e-5\t-5\t9\t{ql}\t0\t9\t{tl}\t10M\n", False, "leaves the query"),
    ("lengths of other databases", "{t}\t100\t0.5\t1e-5\t0\t9\t100000\t0\t9\t{tl}\t10M\n", False, "query length"),
    ("target length of another database", "{t}\t100\t0.5\t1e-5\t0\t9\t{ql}\t0\t9\t100000\t10M\n", False, "target length"),
    ("too few fields", "{t}\t100\t0.5\n", False, "Invalid alignment result record"),
    ("unknown target", "987654321\t100\t0.5\t1e-5\t0\t9\t{ql}\t0\t9\t{tl}\t10M\n", False, "987654321"),
    ("binary junk", "\x01\x02\xff\xfe\t\t\t\t\t\t\t\t\t\t\n", False, "Invalid alignment result record"),
])
def test_convertalis_refuses_records_that_do_not_fit_the_databases(tmp_path, tag, rec, ok, msg):
    """alignment records whose lengths, start positions or backtrace address residues the database entries do not have (a result DB of
    other databases, a damaged file): the reference's convertalis prints whatever lies behind the entry (structureconvertalis.cpp:133-171
    walks the backtrace without looking at the lengths); this module names the record and stops -- checked under ASan + UBSan with
    tools/sanitize_host.sh.  Records that do fit keep working, also without a final newline."""
    import subprocess
    w = str(tmp_path)
    keys = _golden_copy(w)
    lens = {int(l.split()[0]): int(l.split()[2]) - 2 for l in open(os.path.join(w, "db.index"))}
    q, t = keys[0], keys[1]
    assert lens[q] >= 20 and lens[t] >= 20
    b = rec.format(t=t, ql=lens[q], tl=lens[t], qlm5=lens[q] - 5).encode("latin1") + b"\0"
    open(os.path.join(w, "aln"), "wb").write(b)
    open(os.path.join(w, "aln.index"), "w").write(f"{q}\t0\t{len(b)}\n")
    np.array([5], np.int32).tofile(os.path.join(w, "aln.dbtype"))
    exe = os.path.join(os.path.dirname(api.LIB_PATH), "bin", "fsgpu-modules")
    r = subprocess.run([exe, "convertalis", "db", "db", "aln", "out.m8", "--threads", "1", "--format-output", ALL_RECORD_COLUMNS], cwd=w, capture_output=True, timeout=120)
    err = r.stderr.decode("latin1")
    assert r.returncode >= 0, (tag, err[-500:])          # never a signal
    if ok:
        assert r.returncode == 0, (tag, err[-500:])
        line = open(os.path.join(w, "out.m8")).read().split("\t")
        assert len(line) == len(ALL_RECORD_COLUMNS.split(",")) and line[9] == str(lens[q])
    else:
        assert r.returncode != 0 and msg in err, (tag, err[-500:])
    # the default columns need no sequences: the same record is formatted without looking at them, as in the reference
    r = subprocess.run([exe, "convertalis", "db", "db", "aln", "out0.m8", "--threads", "1"], cwd=w, capture_output=True, timeout=120)
    assert r.returncode >= 0


@pytest.mark.parametrize("tag", ["empty index", "data cut in half", "entry beyond the data", "non-numeric index line", "zero-length entry", "no dbtype", "short dbtype", "bytes outside the alphabet"])
def test_makepaddedseqdb_on_damaged_databases(tmp_path, tag):
    """damaged inputs end in an error message or a valid (possibly smaller) output, never in a signal"""
    import subprocess
    w = str(tmp_path)
    _golden_copy(w)
    p = os.path.join(w, "db_ss")
    if tag == "empty index": open(p + ".index", "w").write("")
    elif tag == "data cut in half": d = open(p, "rb").read(); open(p, "wb").write(d[:len(d) // 2])
    elif tag == "entry beyond the data": open(p + ".index", "a").write("999999\t99999999999\t50\n")
    elif tag == "non-numeric index line": open(p + ".index", "a").write("abc\tdef\tghi\n")
    elif tag == "zero-length entry": open(p + ".index", "a").write("999999\t0\t0\n")
    elif tag == "no dbtype": os.remove(p + ".dbtype")
    elif tag == "short dbtype": open(p + ".dbtype", "wb").write(b"\x01")
    else:
        d = bytearray(open(p, "rb").read()); d[10:20] = bytes([200, 255, 0, 1, 2, 127, 128, 129, 10, 0]); open(p, "wb").write(d)
    exe = os.path.join(os.path.dirname(api.LIB_PATH), "bin", "fsgpu-modules")
    r = subprocess.run([exe, "makepaddedseqdb", "db_ss", "pad", "--threads", "1"], cwd=w, capture_output=True, text=True, timeout=120)
    assert r.returncode >= 0, r.stderr[-500:]
    if tag in ("data cut in half", "entry beyond the data", "no dbtype", "short dbtype"):
        assert r.returncode != 0 and r.stderr.strip()
    if r.returncode == 0:
        idx = [l.split() for l in open(os.path.join(w, "pad.index"))]
        size = os.path.getsize(os.path.join(w, "pad"))
        assert all(int(o) + max(int(n) - 2, 0) <= size for _, o, n in idx)      # the padded layout keeps the source's length + 2 in the index


def _write_one_alignment(w, q, t):
    b = f"{t}\t100\t0.5\t1e-5\t0\t9\t50\t0\t9\t60\t10M\n".encode() + b"\0"
    open(os.path.join(w, "aln"), "wb").write(b)
    open(os.path.join(w, "aln.index"), "w").write(f"{q}\t0\t{len(b)}\n")
    np.array([5], np.int32).tofile(os.path.join(w, "aln.dbtype"))


SEQ_COLUMNS = "query,target,qseq,tseq,qheader,theader,qaln,taln,q3di,t3di,q3dialn,t3dialn"


@pytest.mark.parametrize("tag,msg", [
    ("offset 2^64-1 in db.index", "beyond end of data"),
    ("offset near 2^64 in db_h.index", "beyond end of data"),
    ("negative offset in db_ss.index", "beyond end of data"),
    ("header entry without terminator at a page boundary", None),
    ("sequence entry without terminator at a page boundary", None),
    ("alignment entry without terminator, page multiple", "Invalid alignment result record"),
])
def test_database_readers_on_damaged_index_files(tmp_path, tag, msg):
    """index offsets that would wrap the bounds check, and entries that lost their terminator exactly at the end of the mapping: an
    error message or the correct output, never a read past the file (sanitizer run: tools/sanitize_host.sh)"""
    import subprocess
    w = str(tmp_path)
    keys = _golden_copy(w)
    q, t = keys[0], keys[1]
    _write_one_alignment(w, q, t)

    def unterminate(name):
        p = os.path.join(w, name)
        d = open(p, "rb").read()
        lines = [l.split() for l in open(p + ".index")]
        d2 = d + b"A" * ((-len(d)) % 4096 or 4096)
        open(p, "wb").write(d2)
        with open(p + ".index", "w") as f:
            for k, o, n in lines:
                f.write(f"{k}\t{len(d)}\t{len(d2) - len(d)}\n" if int(k) == q else f"{k}\t{o}\t{n}\n")
        return len(d2) - len(d)
    if tag.startswith("offset 2^64-1"): open(os.path.join(w, "db.index"), "a").write("5000\t18446744073709551615\t10\n")
    elif tag.startswith("offset near"): open(os.path.join(w, "db_h.index"), "a").write("5001\t18446744073709551610\t100\n")
    elif tag.startswith("negative"): open(os.path.join(w, "db_ss.index"), "a").write("5002\t-5\t100\n")
    elif tag.startswith("header"): unterminate("db_h")
    elif tag.startswith("sequence"): unterminate("db")
    else:
        rec = f"{t}\t100\t0.5\t1e-5\t0\t9\t50\t0\t9\t60\t10M\n".encode()
        blob = rec * (4096 // len(rec))
        blob += b"1" * (4096 - len(blob))
        open(os.path.join(w, "aln"), "wb").write(blob)
        open(os.path.join(w, "aln.index"), "w").write(f"{q}\t0\t4096\n")
    exe = os.path.join(os.path.dirname(api.LIB_PATH), "bin", "fsgpu-modules")
    r = subprocess.run([exe, "convertalis", "db", "db", "aln", "out.m8", "--threads", "1", "--format-output", SEQ_COLUMNS], cwd=w, capture_output=True, timeout=120)
    assert r.returncode >= 0
    if msg is None:
        assert r.returncode in (0, 1)            # the grown entry may no longer fit the record's lengths: then it is named, not read
        if r.returncode == 0:
            assert len(open(os.path.join(w, "out.m8")).read().split("\t")) == len(SEQ_COLUMNS.split(","))
    else:
        assert r.returncode != 0 and msg in r.stderr.decode("latin1")


@pytest.mark.parametrize("tag,msg", [
    ("intact", None),
    ("data cut in half", "beyond end of data"),
    ("entry count 2^60", "truncated DBR1INDEX"),
    ("entry count 2^64-1", "truncated DBR1INDEX"),
    ("first offset 2^64-1", "beyond the data blob"),
    ("first length 2^32-1", "beyond the data blob"),
    ("index of the index names the wrong blobs", "truncated DBR1INDEX"),
])
def test_sequences_out_of_a_damaged_precomputed_index(tmp_path, tag, msg):
    """`indexdb --index-subset 2` (host only) writes <db>.idx; with the plain sequence DBs gone the modules read the sequences out of it
    (DbReader::openInsideIndex).  Damaged counts / offsets / lengths inside the serialised reader are refused by name."""
    import struct
    import subprocess
    import shutil
    w = str(tmp_path)
    keys = _golden_copy(w)
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scop_v1")
    for e in ("", ".index", ".dbtype"):
        shutil.copy(os.path.join(gold, "db_h" + e), os.path.join(w, "db_ss_h" + e))
    exe = os.path.join(os.path.dirname(api.LIB_PATH), "bin", "fsgpu-modules")
    for name in ("db", "db_ss"):
        r = subprocess.run([exe, "indexdb", name, name, "--index-subset", "2"], cwd=w, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
    expect = None
    _write_one_alignment(w, keys[0], keys[1])
    r = subprocess.run([exe, "convertalis", "db", "db", "aln", "plain.m8", "--threads", "1", "--format-output", SEQ_COLUMNS], cwd=w, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    expect = open(os.path.join(w, "plain.m8")).read()
    for f in ("db", "db.index", "db.dbtype", "db_ss", "db_ss.index", "db_ss.dbtype") + (("db_h", "db_h.index", "db_h.dbtype") if tag == "intact" else ()):
        os.remove(os.path.join(w, f))          # "intact": the headers, too, come out of the index (HDR1INDEX / HDR1DATA)
    for name in ("db", "db_ss"):
        e = {int(l.split()[0]): (int(l.split()[1]), int(l.split()[2])) for l in open(os.path.join(w, name + ".idx.index"))}
        p = os.path.join(w, name + ".idx")
        d = open(p, "rb").read()
        o5 = e[5][0]
        if tag == "data cut in half": d = d[:len(d) // 2]
        elif tag == "entry count 2^60": d = d[:o5] + struct.pack("<Q", 1 << 60) + d[o5 + 8:]
        elif tag == "entry count 2^64-1": d = d[:o5] + struct.pack("<Q", (1 << 64) - 1) + d[o5 + 8:]
        elif tag == "first offset 2^64-1": d = d[:o5 + 36] + struct.pack("<Q", (1 << 64) - 1) + d[o5 + 44:]
        elif tag == "first length 2^32-1": d = d[:o5 + 44] + struct.pack("<I", (1 << 32) - 1) + d[o5 + 48:]
        elif tag.startswith("index of the index"): open(p + ".index", "w").write("5\t0\t10\n6\t0\t10\n")
        open(p, "wb").write(d)
    r = subprocess.run([exe, "convertalis", "db.idx", "db.idx", "aln", "idx.m8", "--threads", "1", "--format-output", SEQ_COLUMNS], cwd=w, capture_output=True, text=True, timeout=120)
    assert r.returncode >= 0
    if msg is None:
        assert r.returncode == 0, r.stderr
        assert open(os.path.join(w, "idx.m8")).read() == expect          # same text from the index as from the plain databases
    else:
        assert r.returncode != 0 and msg in r.stderr, r.stderr


def _manifest_vectors():
    import json
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scop_v1")
    m = json.load(open(os.path.join(gold, "MANIFEST.json")))
    out = []
    for group in ("runs", "runs_with_index", "convert_runs"):
        for name, r in m[group].items():
            out.append((f"{group}:{name}", r["module"], list(r["positional"]), list(r["parameters"])))
    for i, v in enumerate(m["indexdb"]):
        out.append((f"indexdb:{i}", v[0], [], list(v[1:])))
    return out


def test_every_reference_parameter_vector_is_inside_the_value_domains(tmp_path):
    """the complete parameter vectors the reference's workflows handed to its modules (tests/golden/scop_v1/MANIFEST.json, 60+ runs) and the
    edge values of test_modules_vs_reference_binary.py must pass the argument parser -- incl. the reference's per-parameter value patterns
    (`domainError`) -- so without a GPU every device module ends at the DEVICE error, not at an argument error"""
    import subprocess
    w = str(tmp_path)
    _golden_copy(w)
    exe = os.path.join(os.path.dirname(api.LIB_PATH), "bin", "fsgpu-modules")
    vectors = _manifest_vectors()
    edge = {"-e": ["0", "1e-300", "1e+30", "10", "0.001"], "--max-accept": ["1"], "-c": ["1.0", "0.8"], "--min-seq-id": ["1.0"], "--min-aln-len": ["100000"],
            "--alt-ali": ["5"], "--gap-open": ["aa:2,nucl:2", "7"], "--max-seqs": ["1", "100000"], "-s": ["1", "9.5", "7.5"], "--k-score": ["seq:200,prof:200"],
            "--min-ungapped-score": ["0", "255"], "--mask-n-repeat": ["2"], "--comp-bias-corr-scale": ["1.0", "0.15"], "--split-memory-limit": ["0", "12G"]}
    base = next(v for v in vectors if v[1] == "structurealign")
    pbase = next(v for v in vectors if v[1] == "prefilter")
    for k, vals in edge.items():
        for v in vals:
            src = base if k in base[3] else pbase
            par = list(src[3]); par[par.index(k) + 1] = v
            vectors.append((f"edge:{k}={v}", src[1], src[2], par))
    assert len(vectors) > 60
    for tag, module, pos, par in vectors:
        if module == "convertalis" or (module == "indexdb" and "--index-subset" in par and par[par.index("--index-subset") + 1] == "2"):
            continue                                       # host-only: run for real by test_scop_golden / test_indexdb tests
        pos = [p for p in pos] + ([f"out_{abs(hash(tag))}"] if module != "indexdb" else [])
        r = subprocess.run([exe, module] + pos + par, cwd=w, capture_output=True, text=True, timeout=300)
        bad = [m for m in ("Error in", "Unrecognized parameter", "Missing argument", "Invalid boolean", "Duplicate parameter") if m in r.stderr]
        assert not bad, (tag, r.stderr[-400:])


@pytest.mark.parametrize("flag,value,msg", [
    ("--threads", "0", "Error in argument --threads"), ("--threads", "abc", "Error in argument --threads"), ("--threads", "-5", "Error in argument --threads"),
    ("--max-seqs", "0", "Error in argument --max-seqs"), ("--max-seqs", "1.5", "Error in argument --max-seqs"),
    ("-e", "", None), ("-e", "nan", None),                  # the reference's pattern for -e accepts an empty match: anything goes there, too
    ("-c", "1.5", "Error in argument -c"), ("-c", "-0.1", "Error in argument -c"), ("--cov-mode", "6", "Error in argument --cov-mode"),
    ("--alignment-type", "4", "Error in argument --alignment-type"), ("--min-seq-id", "2", "Error in argument --min-seq-id"),
    ("--gap-open", "12abc", "Error in value parsing --gap-open"), ("--gap-open", "aa:3", None), ("--gap-open", "aa:3,nucl:x1", None),
    ("--gap-open", "aa:3,nucl:1x", "Error in value parsing --gap-open"), ("--gap-open", "aa:3,prot:4", "Error in value parsing --gap-open"),
    ("--gap-extend", "99999999999999999999", "Error in value parsing --gap-extend"),
    ("--comp-bias-corr", "2", "Error in argument --comp-bias-corr"), ("--alt-ali", "-1", "Error in argument --alt-ali"),
])
def test_values_outside_the_reference_domains_get_the_reference_message(tmp_path, flag, value, msg):
    """Parameters::parseParameters validates every value against the parameter's pattern before anything runs (Parameters.cpp:1905-2040)
    and ends in "Error in argument <flag>" / "Error in value parsing <flag>"; same here, with the same patterns"""
    import subprocess
    w = str(tmp_path)
    _golden_copy(w)
    exe = os.path.join(os.path.dirname(api.LIB_PATH), "bin", "fsgpu-modules")
    module = ["prefilter", "db", "db", "out"] if flag == "--max-seqs" else ["structurealign", "db", "db", "pref", "out"]
    r = subprocess.run([exe] + module + [flag, value], cwd=w, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 or msg is None               # (a GPU box runs the prefilter; here: no such prefilter DB / no device at the latest)
    if msg is None:
        assert "Error in" not in r.stderr, r.stderr
    else:
        assert msg in r.stderr, r.stderr


@pytest.mark.parametrize("tag,msg", [
    ("server already gone", "GPU server has unexpectedly shut down"),
    ("server already gone, default geometry", "GPU server has unexpectedly shut down"),
    ("block smaller than its header says", "smaller than its header"),
    ("profile area outside the block", "unexpected layout"),
    ("query area inside the header", "unexpected layout"),
    ("results area wraps", "unexpected layout"),
    ("garbage", None),
])
def test_gpuserver_client_checks_the_shared_memory_block(tmp_path, tag, msg):
    """the client side of the gpuserver protocol (no GPU involved: `ungappedprefilter --gpu-server 1`) against hand-made blocks under the
    server's name: a block whose areas do not lie inside it (another program / version under the same name) is refused before anything is
    written into it; a well-formed block of a server that has exited ends with the reference's message"""
    import struct
    import subprocess
    rng = np.random.default_rng(5)
    seqs = [rng.integers(0, 20, size=50).astype(np.uint8) for _ in range(6)]
    src = str(tmp_path / "db_ss")
    dbio.write_seq_db(src, seqs, list(range(6)))
    exe = os.path.join(os.path.dirname(api.LIB_PATH), "bin", "fsgpu-modules")
    name = "fsgpu_test_fake_%d_%d" % (os.getpid(), abs(hash(tag)) % 100000)
    max_len, max_res = (65535, 1000) if "default geometry" in tag else (1001, 300)      # odd: the result area is unaligned, as with the default 65535
    size = 36 + max_len + 16 * max_res + 21 * max_len
    qoff, roff, poff = 36, 36 + max_len, 36 + max_len + 16 * max_res          # the reference's order: query, results, profile
    state, server_exit = 0, 0
    if tag.startswith("server already gone"): server_exit = 1
    elif tag.startswith("profile"): poff = size - 100
    elif tag.startswith("query"): qoff = 8
    elif tag.startswith("results"): roff = 0xfffffff0
    hdr = struct.pack("<IIiB3xIIIII", max_len, max_res, state, server_exit, qoff, 0, roff, 0, poff)
    assert len(hdr) == 36
    blob = hdr + bytes(size - 36)
    if tag.startswith("block smaller"): blob = blob[:size // 2]
    if tag == "garbage": blob = rng.integers(0, 256, size, dtype=np.uint8).tobytes()
    path = "/dev/shm/" + name
    open(path, "wb").write(blob)
    try:
        r = subprocess.run([exe, "ungappedprefilter", src, src, str(tmp_path / "out"), "--gpu-server", "1", "--gpu-server-wait-timeout", "0", "--shm-name", name],
                           capture_output=True, text=True, timeout=60)
    finally:
        os.remove(path)
    assert r.returncode > 0, r.stderr                  # an error exit, never a signal
    if msg:
        assert msg in r.stderr, r.stderr


def test_gpuserver_client_round_trip_against_a_scripted_server(tmp_path):
    """the client's half of the protocol end to end without a GPU: a scripted server (this test, through the same /dev/shm block layout the
    reference's GPUSharedMemory has) answers every READY with a made-up result list; the client must hand over each query's residue codes and
    21 x L profile, take the list, drop what is not above --min-ungapped-score, order by (score desc, key asc), cut at --max-seqs and write
    the prefilter DB in the reference's text format"""
    import mmap
    import struct
    import subprocess
    import threading
    import time
    rng = np.random.default_rng(9)
    lens = [50, 1, 333, 0, 77, 120]
    seqs = [rng.integers(0, 20, size=L).astype(np.uint8) for L in lens]
    keys = [3, 5, 8, 13, 21, 34]
    src = str(tmp_path / "db_ss")
    dbio.write_seq_db(src, seqs, keys)
    exe = os.path.join(os.path.dirname(api.LIB_PATH), "bin", "fsgpu-modules")
    name = "fsgpu_test_script_%d" % os.getpid()
    max_len, max_res = 65535, 10
    qoff, roff = 36, 36 + max_len
    poff = roff + 16 * max_res
    size = poff + 21 * max_len
    path = "/dev/shm/" + name
    with open(path, "wb") as f:
        f.write(struct.pack("<IIiB3xIIIII", max_len, max_res, 0, 0, qoff, 0, roff, 0, poff) + bytes(size - 36))
    fd = os.open(path, os.O_RDWR)
    mm = mmap.mmap(fd, size)
    seen, stop = [], threading.Event()

    def made_up(codes):                  # (target id, score): ids of the 6 targets, two ties, one score at the threshold, one below
        s = int(codes.sum()) % 50
        return [(5, 100 + s), (0, 31), (2, 100 + s), (4, 30), (1, 255), (3, 12)]

    def server():
        while not stop.is_set():
            if struct.unpack_from("<i", mm, 8)[0] != 2:          # READY
                time.sleep(0.0005)
                continue
            L = struct.unpack_from("<I", mm, 20)[0]
            codes = np.frombuffer(mm[qoff:qoff + L], np.uint8).copy()
            prof = np.frombuffer(mm[poff:poff + 21 * L], np.int8).copy().reshape(21, L)
            seen.append((codes, prof))
            res = made_up(codes)
            for i, (tid, sc) in enumerate(res):
                struct.pack_into("<Iiii", mm, roff + 16 * i, tid, sc, 0, 0)
            struct.pack_into("<I", mm, 28, len(res))               # resultLen
            struct.pack_into("<i", mm, 8, 3)                       # DONE
    th = threading.Thread(target=server, daemon=True)
    th.start()
    try:
        r = subprocess.run([exe, "ungappedprefilter", src, src, str(tmp_path / "out"), "--gpu-server", "1", "--gpu-server-wait-timeout", "0", "--shm-name", name,
                            "--max-seqs", "4", "--comp-bias-corr", "0"], capture_output=True, text=True, timeout=120)
    finally:
        stop.set(); th.join(timeout=5)
        mm.close(); os.close(fd); os.remove(path)
    assert r.returncode == 0, r.stderr
    nonempty = [s for s in seqs if len(s)]
    assert len(seen) == len(nonempty)
    for (codes, prof), s in zip(seen, nonempty):
        assert (codes == s).all()                                  # residue codes in the padded alphabet's numbering
        assert prof.shape == (21, len(s)) and len(np.unique(prof[:20], axis=0)) > 1 if len(s) > 5 else True
        for j in range(len(s)):                                    # no bias: a column of the profile is the matrix row of the query residue
            assert (prof[:, j] == prof[:, int(np.flatnonzero(s == s[j])[0])]).all()
    data = open(tmp_path / "out", "rb").read()
    idx = {int(l.split()[0]): (int(l.split()[1]), int(l.split()[2])) for l in open(str(tmp_path / "out") + ".index")}
    assert sorted(idx) == keys
    for k, s in zip(keys, seqs):
        entry = data[idx[k][0]:idx[k][0] + idx[k][1] - 1].decode()
        if len(s) == 0:
            assert entry == ""
            continue
        hits = [(keys[t], sc) for t, sc in made_up(s) if sc > 30 or keys[t] == k]        # identity (same DB) stays whatever its score
        hits.sort(key=lambda h: (-h[1], h[0]))
        assert entry == "".join(f"{t}\t{sc}\t0\n" for t, sc in hits[:4]), (k, entry)
