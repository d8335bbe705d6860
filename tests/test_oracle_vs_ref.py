"""Pins the oracle (oracle/fs_oracle.c) and the host-side product code to the reference's own compiled sources
(oracle/_ref/libfsref.so, built in-container from /root/reference by oracle/Makefile).  Skipped where _ref was not built.
The same comparisons are frozen as committed fixtures in tests/golden/ (test_golden.py) so they also run without _ref."""
import ctypes as C
import os

import numpy as np
import pytest

from foldseek_amd import api, synth
import helpers
from oracle_lib import load_ref, load_oracle, REFSW_DT, REFALN_DT, SW_DT

R = load_ref()
pytestmark = pytest.mark.skipif(R is None, reason="oracle/_ref not built (needs /root/reference at build time)")


def test_matrices_and_bias_match_reference():
    rng = np.random.default_rng(0)
    for which, name in ((0, "MAT3DI"), (1, "BLOSUM62")):
        for bf, sb in ((2.0, 0.0), (2.1, 0.0), (8.0, -0.2), (1.4, 0.0), (0.0, 0.0)):
            sub = np.zeros(21 * 21, np.int16)
            pb = np.zeros(21)
            R.ref_submat(which, bf, sb, sub, pb)
            s2, p2 = helpers.o_submat(name, bf, sb)
            assert (sub == s2).all() and (pb == p2).all()
            for L in (1, 39, 41, 350):
                seq = rng.integers(0, 21, size=L).astype(np.uint8)
                for scale in (0.15, 0.5, 1.0):
                    a = np.zeros(L, np.float32)
                    R.ref_comp_bias(which, bf, sb, seq, L, scale, a)
                    b, _ = helpers.o_round_bias(s2, p2, seq, scale)
                    assert (a == b).all()


def test_ungapped_matches_reference():
    q3, qa = synth.make_queries(3, seed=5)
    db = synth.make_db(1500, (q3, qa), seed=7, homologs_per_query=30, mask_frac=0.02)
    for qi in range(3):
        for cb in (1, 0):
            ref = np.zeros(db.n, np.int32)
            R.ref_ungapped(q3[qi], len(q3[qi]), cb, 0.15, db.data3di, db.offsets[:-1].copy(), db.lengths, db.n, 1, ref)
            assert (ref == helpers.o_ungapped_scores(q3[qi], db, bool(cb))).all()


@pytest.mark.parametrize("atype", [2, 0])
def test_sw_matches_reference(atype):
    q3, qa = synth.make_queries(3, seed=15)
    db = synth.make_db(400, (q3, qa), seed=17, homologs_per_query=40)
    t3 = np.where(db.data3di >= 32, db.data3di - 32, db.data3di).astype(np.uint8)
    for qi in range(3):
        L = len(q3[qi])
        fw = np.zeros(db.n, REFSW_DT)
        rv = np.zeros(db.n, REFSW_DT)
        R.ref_structure_align(qa[qi], q3[qi], L, atype, 1, 0.5, 10, 1, db.dataaa, t3, db.offsets[:-1].copy(), db.lengths, db.n,
                              db.residues, 10.0, 0, 1, fw.ctypes.data, rv.ctypes.data, None, None, 0)
        pAf, p3f, _, _ = helpers.o_align_profiles(qa[qi], q3[qi], atype)
        pAr, p3r, _, _ = helpers.o_align_profiles(qa[qi][::-1].copy(), q3[qi][::-1].copy(), atype)
        for i in range(db.n):
            ta, tt = helpers.target_seqs(db, i)
            for (pA, p3, ref) in ((pAf, p3f, fw[i]), (pAr, p3r, rv[i])):
                w = helpers.o_sw(pA, p3, L, ta, tt)
                assert (w["score"], w["qEnd"], w["dbEnd"], w["word"]) == (ref["score"], ref["qEnd"], ref["dbEnd"], ref["word"])


@pytest.mark.parametrize("go,ge", [(5, 5), (3, 7), (1, 1)])
def test_gap_open_not_above_gap_extend_scores_equal_and_the_reference_aborts_on_the_backtrace(go, ge):
    """gapOpen <= gapExtend: the score / end-position half of the reference (striped SW, lazy-F loop leaving early) equals the oracle's literal
    emulation; as soon as a hit is accepted the reference goes through the block aligner, whose assertion `gaps.open < gaps.extend`
    (scan_block.rs:864-867) ends the process -- so there is no reference answer for the device path to reproduce"""
    import subprocess
    import sys
    q3, qa = synth.make_queries(1, seed=15)
    db = synth.make_db(120, (q3, qa), seed=17, homologs_per_query=20)
    t3 = np.where(db.data3di >= 32, db.data3di - 32, db.data3di).astype(np.uint8)
    L = len(q3[0])
    fw = np.zeros(db.n, REFSW_DT)
    rv = np.zeros(db.n, REFSW_DT)
    R.ref_structure_align(qa[0], q3[0], L, 2, 1, 0.5, go, ge, db.dataaa, t3, db.offsets[:-1].copy(), db.lengths, db.n,
                          db.residues, 10.0, 0, 1, fw.ctypes.data, rv.ctypes.data, None, None, 0)
    pAf, p3f, _, _ = helpers.o_align_profiles(qa[0], q3[0], 2)
    for i in range(db.n):
        ta, tt = helpers.target_seqs(db, i)
        w = helpers.o_sw(pAf, p3f, L, ta, tt, go, ge)
        assert (w["score"], w["qEnd"], w["dbEnd"], w["word"]) == (fw[i]["score"], fw[i]["qEnd"], fw[i]["dbEnd"], fw[i]["word"])
    code = f"""
import sys
sys.path.insert(0, {os.path.dirname(os.path.abspath(__file__))!r}); sys.path.insert(0, {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r})
import numpy as np, oracle_lib
from foldseek_amd import synth
R = oracle_lib.load_ref()
q3, qa = synth.make_queries(1, seed=15)
db = synth.make_db(120, (q3, qa), seed=17, homologs_per_query=20)
t3 = np.where(db.data3di >= 32, db.data3di - 32, db.data3di).astype(np.uint8)
n = db.n
aln = np.zeros(n, oracle_lib.REFALN_DT); cig = np.zeros(1 << 20, np.uint8)
R.ref_structure_align(qa[0], q3[0], len(q3[0]), 2, 1, 0.5, {go}, {ge}, db.dataaa, t3, db.offsets[:-1].copy(), db.lengths, n,
                      db.residues, 10.0, 1, 1, None, None, aln.ctypes.data, cig.ctypes.data, cig.size)
print("survived")
"""
    p = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode != 0 and "survived" not in p.stdout and "open" in p.stderr


def test_rowmajor_recurrence_equals_striped_emulation():
    """the recurrence the HIP kernel implements == the lane-by-lane emulation, on adversarial small-alphabet inputs"""
    O = load_oracle()
    rng = np.random.default_rng(4)
    a = np.zeros(1, SW_DT)
    b = np.zeros(1, SW_DT)
    for _ in range(1500):
        Lq, Lt = int(rng.integers(1, 200)), int(rng.integers(1, 200))
        lo, hi = int(rng.integers(-20, -1)), int(rng.integers(1, 12))
        pA = rng.integers(lo, hi + 1, size=21 * Lq).astype(np.int16)
        p3 = rng.integers(lo, hi + 1, size=21 * Lq).astype(np.int16)
        tA = rng.integers(0, 21, size=Lt).astype(np.uint8)
        t3 = rng.integers(0, 21, size=Lt).astype(np.uint8)
        go = int(rng.integers(2, 14))
        ge = int(rng.integers(1, go))
        for lanes, sat in ((16, 1), (8, 0)):
            O.fso_sw_pass(pA, p3, Lq, tA, t3, Lt, go, ge, lanes, sat, a.ctypes.data)
            O.fso_sw_rowmajor(pA, p3, Lq, tA, t3, Lt, go, ge, lanes, sat, b.ctypes.data)
            assert tuple(a[0]) == tuple(b[0])


def test_evalue_network_matches_reference():
    O = load_oracle()
    import os
    nn = np.fromfile(os.path.join(os.path.dirname(api.LIB_PATH), "data", "evalue_nn.bin"), dtype=np.uint8)
    rng = np.random.default_rng(9)
    ev = api.Evaluer(35_000_000)
    for _ in range(100):
        L = int(rng.integers(20, 2000))
        q = rng.integers(0, 21, size=L).astype(np.uint8)
        c, d = C.c_double(), C.c_double()
        R.ref_mu_lambda(q, L, 35_000_000, C.byref(c), C.byref(d))
        a, b = C.c_double(), C.c_double()
        O.fso_predict_mu_lambda(nn, q, L, 21, C.byref(a), C.byref(b))
        assert (a.value, b.value) == (c.value, d.value) == ev.mu_lambda(q)
        for sc in (-5, 20, 33, 70, 150, 900):
            e = R.ref_evalue_corr(sc, c.value, d.value, 35_000_000)
            assert e == O.fso_evalue_corr(sc, c.value, d.value, np.log(35_000_000.0)) == ev.evalue_corr(sc, c.value, d.value)


@pytest.mark.parametrize("atype", [2, 0])
def test_block_backtrace_matches_reference(atype):
    """host start position / CIGAR / identities == the reference's alignStartPosBacktraceBlock on every accepted hit"""
    q3, qa = synth.make_queries(3, seed=25)
    db = synth.make_db(600, (q3, qa), seed=27, homologs_per_query=50)
    t3 = np.where(db.data3di >= 32, db.data3di - 32, db.data3di).astype(np.uint8)
    mAA = api.Matrix(1, 1.4 if atype == 2 else 0.0)
    m3 = api.Matrix(0, 2.1)
    naccepted = 0
    for qi in range(3):
        L = len(q3[qi])
        aln = np.zeros(db.n, REFALN_DT)
        cig = C.create_string_buffer(4_000_000)
        R.ref_structure_align(qa[qi], q3[qi], L, atype, 1, 0.5, 10, 1, db.dataaa, t3, db.offsets[:-1].copy(), db.lengths, db.n,
                              db.residues, 10.0, 1, 1, None, None, aln.ctypes.data, cig, 4_000_000)
        cigs = cig.value.decode().split("\n")
        _, _, cbA, cbS = api.align_profiles(mAA, m3, qa[qi], q3[qi], True, 0.5)
        for i in np.flatnonzero(aln["status"] == 0):
            r = aln[i]
            ta, tt = helpers.target_seqs(db, int(i))
            ok, qs, ds, ident, bt = api.block_backtrace(mAA, m3, qa[qi], q3[qi], cbA, cbS, ta, tt, r["qEnd"], r["dbEnd"], r["fwdScore"])
            assert ok == (r["qStart"] >= 0)
            assert (qs, ds) == (r["qStart"], r["dbStart"])
            assert bt == cigs[i]
            if ok:
                assert ident == r["identicalAA"]
            naccepted += 1
    assert naccepted > 100
