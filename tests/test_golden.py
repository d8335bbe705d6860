"""Committed golden vectors (tests/golden/hotpath_v1.npz, generated from the reference's own compiled code by
tests/golden/make_golden.py).  CPU part: oracle and host code against the fixture.  GPU part (-m gpu): the device
pipeline through the C ABI against the fixture -- complete result lines included."""
import ctypes as C
import os
import numpy as np
import pytest

from foldseek_amd import api, synth
import helpers
from oracle_lib import load_oracle

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hotpath_v1.npz"))


def _queries():
    lens = G["q_lens"]
    o = np.concatenate([[0], np.cumsum(lens)])
    q3 = [np.ascontiguousarray(G["q3"][o[i]:o[i + 1]]) for i in range(len(lens))]
    qa = [np.ascontiguousarray(G["qa"][o[i]:o[i + 1]]) for i in range(len(lens))]
    return q3, qa


def _db():
    return synth.PaddedDB(np.ascontiguousarray(G["db_data3di"]), np.ascontiguousarray(G["db_dataaa"]), G["db_offsets"], G["db_lengths"])


def test_oracle_matrices_against_golden():
    for name, key in (("MAT3DI", "mat3di"), ("BLOSUM62", "blosum62")):
        for bf in (2.0, 2.1, 1.4, 0.0):
            sub, pb = helpers.o_submat(name, bf)
            assert (sub == G[f"sub_{key}_{bf}"]).all() and (pb == G[f"pback_{key}_{bf}"]).all()
            m = api.Matrix(0 if key == "mat3di" else 1, bf)
            assert (m.scores().ravel() == G[f"sub_{key}_{bf}"]).all()


def test_oracle_ungapped_against_golden():
    q3, _ = _queries()
    db = _db()
    for qi in range(len(q3)):
        assert (helpers.o_ungapped_scores(q3[qi], db, True) == G["ungapped"][qi]).all()
        assert (helpers.o_ungapped_scores(q3[qi], db, False) == G["ungapped_nocb"][qi]).all()


@pytest.mark.parametrize("atype", [2, 0])
def test_oracle_sw_against_golden(atype):
    q3, qa = _queries()
    db = _db()
    for qi in (0, 2, 4):
        L = len(q3[qi])
        pAf, p3f, _, _ = helpers.o_align_profiles(qa[qi], q3[qi], atype)
        pAr, p3r, _, _ = helpers.o_align_profiles(qa[qi][::-1].copy(), q3[qi][::-1].copy(), atype)
        for i in range(0, db.n, 3):
            ta, tt = helpers.target_seqs(db, i)
            for (pA, p3, ref) in ((pAf, p3f, G[f"sw_fwd_t{atype}"][qi][i]), (pAr, p3r, G[f"sw_rev_t{atype}"][qi][i])):
                w = helpers.o_sw(pA, p3, L, ta, tt)
                assert (w["score"], w["qEnd"], w["dbEnd"], w["word"]) == (ref["score"], ref["qEnd"], ref["dbEnd"], ref["word"])


def test_host_evalue_and_backtrace_against_golden():
    q3, qa = _queries()
    db = _db()
    ev = api.Evaluer(db.residues)
    for atype in (2, 0):
        mAA = api.Matrix(1, 1.4 if atype == 2 else 0.0)
        m3 = api.Matrix(0, 2.1)
        for qi in range(len(q3)):
            lam, mu = ev.mu_lambda(q3[qi])
            assert (lam, mu) == tuple(G["mu_lambda"][qi])
            _, _, cbA, cbS = api.align_profiles(mAA, m3, qa[qi], q3[qi], True, 0.5)
            cigs = str(G[f"cigars_t{atype}"][qi]).split("\n")
            aln = G[f"aln_t{atype}"][qi]
            for i in np.flatnonzero(aln["status"] == 0):
                r = aln[i]
                assert ev.evalue_corr(r["score"], lam, mu) == r["evalue"]
                ta, tt = helpers.target_seqs(db, int(i))
                ok, qs, ds, ident, bt = api.block_backtrace(mAA, m3, qa[qi], q3[qi], cbA, cbS, ta, tt, r["qEnd"], r["dbEnd"], r["fwdScore"])
                assert (qs, ds, bt) == (r["qStart"], r["dbStart"], cigs[i])
                if ok:
                    assert ident == r["identicalAA"]


# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("atype", [2, 0])
def test_device_pipeline_against_golden(atype):
    q3, qa = _queries()
    db = _db()
    ctx = api.Context(0)
    ctx.load_db(db)
    par = api.default_params()
    par.alignmentType = atype
    par.maxResListLen = db.n
    par.minDiagScoreThr = -1            # keep every target: the fixture holds all pairs
    search = api.Search(ctx, par)
    for qi in range(len(q3)):
        hits = search.prefilter(q3[qi])
        assert (ctx.gapless_scores().astype(np.int32) == G["ungapped"][qi]).all()
        sel = helpers.o_prefilter_select(G["ungapped"][qi], -1, -1, db.n)
        assert (hits["id"] == sel["key"]).all() and (hits["score"] == sel["score"]).all()
        ids = np.arange(db.n, dtype=np.uint32)
        res, bts = search.align(qa[qi], q3[qi], ids, with_backtrace=True)
        fwd, rev = search.last_sw(db.n)
        gf, gr = G[f"sw_fwd_t{atype}"][qi], G[f"sw_rev_t{atype}"][qi]
        for f in ("score", "qEnd", "dbEnd", "word"):
            assert (fwd[f] == gf[f]).all() and (rev[f] == gr[f]).all(), f
        aln = G[f"aln_t{atype}"][qi]
        cigs = str(G[f"cigars_t{atype}"][qi]).split("\n")
        acc = np.flatnonzero(aln["status"] == 0)
        assert len(res) == len(acc)
        by_key = {int(r["dbKey"]): (r, bt) for r, bt in zip(res, bts)}
        for i in acc:
            g = aln[i]
            r, bt = by_key[int(i)]
            assert (r["score"], r["qStartPos"], r["qEndPos"], r["dbStartPos"], r["dbEndPos"], r["alnLength"]) == \
                   (g["score"], g["qStart"], g["qEnd"], g["dbStart"], g["dbEnd"], g["alnLen"])
            assert r["eval"] == g["evalue"] and r["seqId"] == g["seqId"] and r["qcov"] == g["qCov"] and r["dbcov"] == g["tCov"]
            assert bt == cigs[i]
        # result order: Matcher::compareHits
        keys = [(r["eval"], -r["score"], r["dbLen"], r["dbKey"]) for r in res]
        assert keys == sorted(keys)
    search.close()
    ctx.close()
