"""Shared test helpers: the oracle-side computation of what the product path should return."""
import numpy as np
import ctypes as C
from oracle_lib import load_oracle, SW_DT, HIT_DT
import fsparams

_O = None
_P = None


def oracle():
    global _O, _P
    if _O is None:
        _O = load_oracle()
        _P = fsparams.load()
    return _O


def o_submat(name, bit_factor, score_bias=0.0):
    O = oracle()
    p = _P[name]
    sub = np.zeros(21 * 21, np.int16)
    pb = np.zeros(21)
    O.fso_submat_build(np.ascontiguousarray(p["score"].ravel()), p["back"], p["lam"], 21, bit_factor, score_bias, sub, pb)
    return sub, pb


def o_round_bias(sub, pb, seq, scale):
    O = oracle()
    L = len(seq)
    cbf = np.zeros(L, np.float32)
    O.fso_comp_bias(sub, pb, 21, np.ascontiguousarray(seq, np.uint8), L, scale, cbf)
    cb = np.zeros(L, np.int8)
    O.fso_round_bias(cbf, L, cb)
    return cbf, cb


def o_ungapped_scores(q3di, db, comp_bias=True, scale=0.15):
    """oracle scores of one query against every target of a PaddedDB (masked residues -> X)."""
    O = oracle()
    sub, pb = o_submat("MAT3DI", 2.0)
    tiny = sub.astype(np.int8)
    q = np.ascontiguousarray(q3di, np.uint8)
    cb = o_round_bias(sub, pb, q, scale)[1] if comp_bias else np.zeros(len(q), np.int8)
    out = np.zeros(db.n, np.int32)
    for i in range(db.n):
        raw = db.data3di[db.offsets[i]:db.offsets[i] + db.lengths[i]]
        t = np.ascontiguousarray(np.where(raw >= 32, 20, raw).astype(np.uint8))
        out[i] = O.fso_ungapped_score(q, len(q), tiny, 21, cb, t, len(t))
    return out


def o_prefilter_select(scores, min_score=30, identity=-1, max_res=1000):
    O = oracle()
    keys = np.arange(len(scores), dtype=np.uint32)
    out = np.zeros(max_res, HIT_DT)
    n = O.fso_prefilter_select(np.ascontiguousarray(scores, np.int32), keys, len(scores), min_score, identity, max_res,
                               out.ctypes.data)
    return out[:n]


def o_align_profiles(qAA, q3Di, alignment_type=2, comp_bias=True, scale=0.5):
    O = oracle()
    s3, _ = o_submat("MAT3DI", 2.1)
    sa, pa = o_submat("BLOSUM62", 1.4 if alignment_type == 2 else 0.0)
    L = len(q3Di)
    qa = np.ascontiguousarray(qAA, np.uint8)
    q3 = np.ascontiguousarray(q3Di, np.uint8)
    if comp_bias:
        cba = o_round_bias(sa, pa, qa, 1.0)[1]
        cbs = o_round_bias(sa, pa, q3, scale)[1]      # sic: against the AA matrix
    else:
        cba = np.zeros(L, np.int8)
        cbs = np.zeros(L, np.int8)
    pA = np.zeros(21 * L, np.int16)
    p3 = np.zeros(21 * L, np.int16)
    O.fso_sw_profiles(qa, q3, L, sa.astype(np.int8), s3.astype(np.int8), 21, cba, cbs, pA, p3)
    return pA, p3, cba, cbs


def o_sw(pA, p3, L, tAA, t3Di, go=10, ge=1):
    O = oracle()
    res = np.zeros(1, SW_DT)
    O.fso_sw_score_endpos(np.ascontiguousarray(pA).ravel(), np.ascontiguousarray(p3).ravel(), L,
                          np.ascontiguousarray(tAA, np.uint8), np.ascontiguousarray(t3Di, np.uint8), len(t3Di), go, ge,
                          res.ctypes.data)
    return res[0].copy()


def target_seqs(db, i):
    o, l = db.offsets[i], db.lengths[i]
    t3 = db.data3di[o:o + l]
    t3 = np.where(t3 >= 32, t3 - 32, t3).astype(np.uint8)
    ta = db.dataaa[o:o + l]
    ta = np.where(ta >= 32, ta - 32, ta).astype(np.uint8)
    return np.ascontiguousarray(ta), np.ascontiguousarray(t3)
