"""`indexdb` / `createindex`: the precomputed index this repository writes -- sequence / header DBs, masked sequence
lookup, the k-mer table built on the device and renumbered to the reference's k-mer order, the extended 2- and 3-mer matrices -- against the
one the reference BINARY (oracle/_ref_full/bin/foldseek, built by oracle/build_ref_full.sh, travels to the GPU box) writes for the same DB
with the same parameter string, entry by entry (tests/idx_compare.py), and FUNCTIONALLY: the reference's CPU prefilter run on OUR index
returns what it returns on its own."""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

from foldseek_amd import dbio, synth
from idx_compare import compare
from test_scop_golden import BIN, GOLD, read_db

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FS = os.path.join(ROOT, "oracle", "_ref_full", "bin", "foldseek")
MANIFEST = json.load(open(os.path.join(GOLD, "MANIFEST.json")))
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not os.path.exists(FS), reason="oracle/_ref_full/bin/foldseek not built")]


def _run(cmd, cwd):
    r = subprocess.run(cmd, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, " ".join(cmd[:4]) + "\n" + r.stdout[-3000:]


def _both(w, cmd):
    """runs one frozen indexdb command with both binaries, each on its own copy of the databases"""
    for side, exe in (("ref", FS), ("mine", BIN)):
        _run([exe] + cmd, w / side)


def _stage(src, w, names):
    for side in ("ref", "mine"):
        os.makedirs(w / side, exist_ok=True)
        for n in names:
            for ext in ("", ".index", ".dbtype"):
                if os.path.exists(os.path.join(src, n + ext)):
                    shutil.copy(os.path.join(src, n + ext), w / side / (n + ext))


def test_indexes_of_the_scop_set_equal_the_reference_binarys(tmp_path):
    """the two indexdb calls of F/data/structureindex.sh with the frozen parameter strings (tests/golden/scop_v1/MANIFEST.json): <db>.idx without
    k-mer table, <db>_ss.idx with it (900 MB: 384 MB of 3-mer matrix, 512 MB of offsets)"""
    w = tmp_path
    _stage(GOLD, w, ["db", "db_h", "db_ss"])
    for side in ("ref", "mine"):
        for ext in ("", ".index", ".dbtype"):
            os.symlink("db_h" + ext, w / side / ("db_ss_h" + ext))
    for cmd in MANIFEST["indexdb"]:
        _both(w, cmd)
    for name in ("db.idx", "db_ss.idx"):
        assert compare(str(w / "ref" / name), str(w / "mine" / name)) == [], name
    # the reference's CPU k-mer prefilter on our index == on its own (and == the frozen result DB)
    run = MANIFEST["runs_with_index"]["pref_kmer_idx"]
    for side in ("ref", "mine"):
        for f in os.listdir(w / side):
            if f.startswith("db_ss") and ".idx" not in f and not os.path.islink(w / side / f):
                shutil.copy(w / side / f, w / side / ("q" + f))
        _run([FS, "prefilter", "qdb_ss", "db_ss.idx", "pref"] + run["parameters"], w / side)
    assert read_db(str(w / "ref" / "pref")) == read_db(str(w / "mine" / "pref")) == read_db(os.path.join(GOLD, "pref_kmer_idx"))


def test_index_of_a_masked_3000_target_database_equals_the_reference_binarys(tmp_path):
    """3000 synthetic targets of 20..1200 residues with soft-masked (lower-case) stretches and homopolymer runs: lower-case and repeat masking
    in the lookup, 10^5..10^6 table entries; then `createindex` (both calls + header links) against the same two reference calls"""
    w = tmp_path
    q3, qa = synth.make_queries(8, seed=77, mean_len=260, lo=40, hi=600)
    db = synth.make_db(3000, (q3, qa), seed=78, homologs_per_query=10, mean_len=240, lo=20, hi=1200, mask_frac=0.05)
    keys = (np.arange(db.n) * 3 + 5).astype(np.uint32)
    seqs3 = [db.seq(i, "3di", unmask=True) for i in range(db.n)]
    for i in range(0, db.n, 97):                     # homopolymer runs of 7+ (masked by --mask-n-repeat 6) and of exactly 6 (kept)
        s = seqs3[i].copy()
        if len(s) > 40:
            s[5:13] = s[5]; s[20:26] = s[20]
            seqs3[i] = s
    masks = [db.data3di[db.offsets[i]:db.offsets[i] + db.lengths[i]] >= 32 for i in range(db.n)]
    seqsa = [db.seq(i, "aa") for i in range(db.n)]
    os.makedirs(w / "src")
    dbio.write_seq_db(str(w / "src" / "t"), seqsa, keys)
    dbio.write_seq_db(str(w / "src" / "t_ss"), seqs3, keys, masks)
    with open(w / "src" / "t_h", "wb") as f, open(w / "src" / "t_h.index", "w") as fi:
        off = 0
        for k in keys:
            b = f"s{k} synthetic".encode() + b"\n\0"
            f.write(b); fi.write(f"{k}\t{off}\t{len(b)}\n"); off += len(b)
    np.array([12], np.int32).tofile(str(w / "src" / "t_h.dbtype"))
    _stage(str(w / "src"), w, ["t", "t_h", "t_ss"])
    par = MANIFEST["indexdb"][1][3:]
    par = par[:par.index("--index-dbsuffix")]
    par[par.index("--threads") + 1] = "8"
    for ext in ("", ".index", ".dbtype"):
        os.symlink("t_h" + ext, w / "ref" / ("t_ss_h" + ext))
    _run([FS, "indexdb", "t", "t"] + par + ["--index-subset", "2"], w / "ref")
    _run([FS, "indexdb", "t_ss", "t_ss"] + par + ["--index-dbsuffix", "_ss", "--index-subset", "5"], w / "ref")
    # a C-alpha database next to it: structureindex.sh appends it to <db>.idx under the keys 500 / 501 (appenddbtoindex)
    rng = np.random.default_rng(5)
    for side in ("ref", "mine"):
        with open(w / side / "t_ca", "wb") as f, open(w / side / "t_ca.index", "w") as fi:
            off = 0
            r2 = np.random.default_rng(6)
            for k, L in zip(keys, db.lengths):
                b = r2.integers(1, 255, int(L) * 3 // 2 + 8).astype(np.uint8).tobytes() + b"\0"
                f.write(b); fi.write(f"{k}\t{off}\t{len(b)}\n"); off += len(b)
        np.array([101], np.int32).tofile(str(w / side / "t_ca.dbtype"))
    _run([FS, "appenddbtoindex", "t_ca", "t.idx", "--id-list", "500", "-v", "1"], w / "ref")
    os.makedirs(w / "mine" / "tmp")
    _run([BIN, "createindex", "t", "tmp"] + par, w / "mine")
    assert os.path.islink(w / "mine" / "t_ss_h.index")
    for name in ("t.idx", "t_ss.idx"):
        assert compare(str(w / "ref" / name), str(w / "mine" / name)) == [], name
    ents = [l.split() for l in open(w / "mine" / "t_ss.idx.index")]
    assert int([e for e in ents if e[0] == "9"][0][2]) > 6 * 100000           # ENTRIES: > 10^5 six-byte records
