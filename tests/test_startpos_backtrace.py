"""alignStartPosBacktrace (reverse striped pass + banded_sw + computerBacktrace,
F/src/commons/StructureSmithWaterman.cpp:540-773, 1723-1957) against the REFERENCE'S OWN compiled function
(oracle/_ref/libfsref.so: ref_structure_startpos calls StructureSmithWaterman::alignStartPosBacktrace<PROFILE>).

CPU part: the host banded trace-back (fshost_banded_backtrace) given the reference's start cell -> identical CIGAR and identity count.
GPU part: the whole routine (device reverse pass on reversed prefixes + host trace-back) -> identical start cell and CIGAR."""
import numpy as np
import pytest

import helpers
import oracle_lib
from foldseek_amd import api, synth


def _pairs(seed, nq=3, per=14):
    rng = np.random.default_rng(seed)
    q3, qa = synth.make_queries(nq, seed=seed, lo=60, hi=330)
    t3, ta = [], []
    for i in range(nq):
        for k in range(per):
            a, b = synth._mutate(rng, q3[i], qa[i], 0.15 + 0.04 * k, 0.10)
            pre, suf = int(rng.integers(0, 30)), int(rng.integers(0, 30))
            a = np.concatenate([rng.integers(0, 20, pre).astype(np.uint8), a, rng.integers(0, 20, suf).astype(np.uint8)])
            b = np.concatenate([rng.integers(0, 20, pre).astype(np.uint8), b, rng.integers(0, 20, suf).astype(np.uint8)])
            t3.append(a); ta.append(b)
    return q3, qa, t3, ta, per


def _ref_startpos(ref, qa, q3, atype, tas, t3s):
    lens = np.array([len(t) for t in t3s], np.int32)
    off = np.zeros(len(lens), np.int64); off[1:] = np.cumsum(lens)[:-1]
    catA, cat3 = np.concatenate(tas).astype(np.uint8), np.concatenate(t3s).astype(np.uint8)
    out = np.zeros(len(lens) * 4, np.int32)
    cig = np.zeros(1 << 20, np.uint8)
    ref.ref_structure_startpos(qa, q3, len(q3), atype, 1, 0.5, 10, 1, catA, cat3, off, lens, len(lens), out, cig.ctypes.data, cig.size)
    # forward end positions / scores of the same pairs (alignScoreEndPos)
    fwd = np.zeros(len(lens), oracle_lib.REFSW_DT)
    ref.ref_structure_align(qa, q3, len(q3), atype, 1, 0.5, 10, 1, catA, cat3, off, lens, len(lens), 10 ** 9, 1e300, 0, 1, fwd.ctypes.data, None, None, None, 0)
    return out.reshape(-1, 4), cig.tobytes().split(b"\0")[0].decode().split("\n"), fwd


@pytest.mark.parametrize("atype", [2, 0])
def test_banded_backtrace_equals_reference_given_its_start_cell(atype):
    ref = oracle_lib.load_ref()
    if ref is None or not hasattr(ref, "ref_structure_startpos"):
        pytest.skip("oracle/_ref not built")
    q3, qa, t3, ta, per = _pairs(5 + atype)
    mA, m3 = api.Matrix(1, 1.4 if atype == 2 else 0.0, 0.0), api.Matrix(0, 2.1, 0.0)
    checked = gapped = 0
    for qi in range(len(q3)):
        tas, t3s = ta[qi * per:(qi + 1) * per], t3[qi * per:(qi + 1) * per]
        out, cigs, fwd = _ref_startpos(ref, qa[qi], q3[qi], atype, tas, t3s)
        _, _, cbA, cbS = api.align_profiles(mA, m3, qa[qi], q3[qi], comp_bias=True, scale=0.5)
        for k in range(per):
            if out[k][3] != 0 or fwd[k]["score"] < 20:
                continue
            ok, ids, bt = api.banded_backtrace(mA, m3, qa[qi], q3[qi], cbA, cbS, tas[k], t3s[k], out[k][0], fwd[k]["qEnd"], out[k][1], fwd[k]["dbEnd"], fwd[k]["score"])
            assert ok and bt == cigs[k] and ids == out[k][2], (qi, k, bt[:60], cigs[k][:60])
            checked += 1
            gapped += ("I" in bt) or ("D" in bt)
    assert checked >= 30 and gapped >= 10


@pytest.mark.gpu
@pytest.mark.parametrize("atype", [2, 0])
def test_startpos_backtrace_equals_reference(atype):
    ref = oracle_lib.load_ref()
    if ref is None or not hasattr(ref, "ref_structure_startpos"):
        pytest.skip("oracle/_ref not built")
    q3, qa, t3, ta, per = _pairs(9 + atype)
    # the targets as a resident DB (ids in the order they were made)
    order = np.argsort([len(t) for t in t3], kind="stable")
    lens = np.array([len(t3[i]) for i in order], np.int32)
    offsets = np.zeros(len(order) + 1, np.int64); offsets[1:] = np.cumsum((lens + 3) // 4 * 4)
    d3 = np.full(int(offsets[-1]), 20, np.uint8); da = np.full(int(offsets[-1]), 20, np.uint8)
    slot = {}
    for new, old in enumerate(order):
        d3[offsets[new]:offsets[new] + lens[new]] = t3[old]; da[offsets[new]:offsets[new] + lens[new]] = ta[old]
        slot[int(old)] = new
    db = synth.PaddedDB(d3, da, offsets, lens)
    ctx = api.Context(0)
    ctx.load_db(db)
    par = api.default_params()
    par.alignmentType = atype
    s = api.Search(ctx, par)
    checked = 0
    for qi in range(len(q3)):
        tas, t3s = ta[qi * per:(qi + 1) * per], t3[qi * per:(qi + 1) * per]
        out, cigs, fwd = _ref_startpos(ref, qa[qi], q3[qi], atype, tas, t3s)
        for k in range(per):
            if out[k][3] != 0 or fwd[k]["score"] < 20:
                continue
            ok, qs, ds, ids, bt = s.startpos_backtrace(qa[qi], q3[qi], slot[qi * per + k], fwd[k]["qEnd"], fwd[k]["dbEnd"], fwd[k]["score"])
            assert ok, (qi, k)
            assert (qs, ds, ids, bt) == (out[k][0], out[k][1], out[k][2], cigs[k]), (qi, k)
            checked += 1
    assert checked >= 30
    s.close()
    ctx.close()
