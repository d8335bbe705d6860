"""k_sw3 (fsgpu_sw_multi_dir_c: compact queries, images built on the device, 16 / 32 / 64 lanes per target pair, target codes from an LDS ring)
against k_sw (fsgpu_sw_batch: word profiles from the host, one pair per wave, both directions) and the C oracle."""
import numpy as np
import pytest

import helpers
from foldseek_amd import api, synth

pytestmark = pytest.mark.gpu


def _profiles(mat, codes, cb, reverse=False):
    c = codes[::-1] if reverse else codes
    return (mat[:, c].astype(np.int16) + cb.astype(np.int16)[None, :]).astype(np.int16)


def _tiny(m):
    return np.ascontiguousarray(np.array(m.scores()).reshape(21, 21).astype(np.int8))


def _same(got, want, what):
    for fld in ("score", "qEnd", "dbEnd"):
        assert (got[fld] == want[fld]).all(), (what, fld, np.flatnonzero(got[fld] != want[fld])[:5], got[fld][:4], want[fld][:4])


@pytest.fixture(scope="module")
def env():
    rng = np.random.default_rng(77)
    seeds3 = [rng.choice(20, size=L).astype(np.uint8) for L in (60, 350, 700, 1000)]
    seedsa = [rng.choice(20, size=len(s)).astype(np.uint8) for s in seeds3]
    db = synth.make_db(1200, (seeds3, seedsa), seed=91, homologs_per_query=12, lo=1, hi=1900, mask_frac=0.02)
    ctx = api.Context(0)
    ctx.load_db(db)
    yield rng, db, ctx, (seeds3, seedsa)
    ctx.close()


# every (HL, R) class: 32 lanes x R = 1..16 rows (L <= 512), 64 lanes x R = 9..16 (L <= 1024), and one query beyond (profile-based path)
LENGTHS = [1, 20, 33, 64, 65, 97, 129, 161, 193, 225, 257, 289, 321, 350, 353, 385, 417, 449, 481, 512,
           513, 577, 641, 705, 769, 833, 897, 961, 1024, 1100]
PAIRS = [1, 2, 3, 4, 5, 7, 9, 16, 17, 33, 64, 31, 120, 250, 6, 11, 13, 66, 15, 130, 1, 3, 5, 8, 19, 34, 4, 2, 70, 9]


# which lanes-per-pair shape a pair takes is a launch decision (fsgpu.hip sw3MultiImpl): the automatic rule (lists of <= 16 pairs with 64 lanes, the
# 16-lane shape only in calls of >= 100 000 pairs), the 16-lane shape forced for targets of up to 512 / 896 columns, and the short-list rule off --
# every (shape, rows-per-lane) class must give the per-pair kernel's records
SHAPES = {"auto": {}, "mid512": {"FSGPU_SW3_MID": "512"}, "mid896_noshort": {"FSGPU_SW3_MID": "896", "FSGPU_SW3_SHORT": "0"},
          "no16_noshort": {"FSGPU_SW3_MID": "0", "FSGPU_SW3_SHORT": "0"}}


@pytest.fixture(params=list(SHAPES))
def shape(request, monkeypatch):
    for k in ("FSGPU_SW3_MID", "FSGPU_SW3_SHORT"):
        monkeypatch.delenv(k, raising=False)
    for k, v in SHAPES[request.param].items():
        monkeypatch.setenv(k, v)
    return request.param


@pytest.mark.parametrize("atype", [0, 2])
def test_sw3_every_class_equals_the_per_pair_kernel(env, atype, shape):
    rng, db, ctx, seeds = env
    rng = np.random.default_rng(500 + atype)
    mA, m3 = api.Matrix(1, 1.4 if atype == 2 else 0.0), api.Matrix(0, 2.1)
    t3, tA = _tiny(m3), _tiny(mA)
    use_aa = atype == 2
    queries, want_f, want_r = [], [], []
    for i, L in enumerate(LENGTHS):
        related = []
        if i % 3 == 1 and L >= 20:       # a query with relatives among its pairs: the first L residues of a DB entry, a fifth of them redrawn
            cand = np.flatnonzero(db.lengths >= L)
            src = int(cand[rng.integers(0, len(cand))])
            qa, q3 = (x[:L].copy() for x in helpers.target_seqs(db, src))
            redraw = rng.random(L) < 0.2
            q3[redraw] = rng.choice(20, size=int(redraw.sum())); qa[redraw] = rng.choice(20, size=int(redraw.sum()))
            related = [src]
        else:
            q3, qa = rng.choice(20, size=L).astype(np.uint8), rng.choice(20, size=L).astype(np.uint8)
        if L > 4:
            q3[rng.integers(0, L)] = 20      # an X
        _, _, cba_f, cb3_f = api.align_profiles(mA, m3, qa, q3, True, 0.5)
        _, _, cba_r, cb3_r = api.align_profiles(mA, m3, qa[::-1].copy(), q3[::-1].copy(), True, 0.5)
        n = PAIRS[i]
        ids = rng.choice(db.n, size=n, replace=False).astype(np.uint32)
        if n >= 4:
            ids[:4] = [0, db.n - 1, 1, db.n - 2]          # shortest and longest targets of the DB in one wave
        if related and related[0] not in ids:
            ids[-1] = related[0]
        queries.append((qa if use_aa else None, q3, cba_f if use_aa else None, cb3_f, cba_r if use_aa else None, cb3_r, ids))
        p3f, p3r = _profiles(t3, q3, cb3_f), _profiles(t3, q3, cb3_r, True)
        pAf, pAr = (_profiles(tA, qa, cba_f), _profiles(tA, qa, cba_r, True)) if use_aa else (None, None)
        f, r = ctx.sw_batch(pAf, p3f, pAr, p3r, ids)
        want_f.append(f); want_r.append(r)
    got_f = ctx.sw_multi_dir_c(t3, tA if use_aa else None, queries, 0)
    for i, L in enumerate(LENGTHS):
        _same(got_f[i], want_f[i], (atype, "fwd", L))
    got_r = ctx.sw_multi_dir_c(t3, tA if use_aa else None, queries, 1)          # finds the forward call's images in place
    for i, L in enumerate(LENGTHS):
        _same(got_r[i], want_r[i], (atype, "rev", L))
    assert sum(int((w["score"] > 100).sum()) for w in want_f) >= 8, "related pairs must be among the pairs"
    # the C oracle on a few pairs of three classes
    for i in (4, 13, 23):
        pA, p3 = helpers.o_align_profiles(queries[i][0] if use_aa else np.zeros(LENGTHS[i], np.uint8), queries[i][1], atype)[:2]
        for k in range(min(5, PAIRS[i])):
            ta, tt = helpers.target_seqs(db, int(queries[i][6][k]))
            w = helpers.o_sw(pA, p3, LENGTHS[i], ta, tt)
            assert (int(got_f[i][k]["score"]), int(got_f[i][k]["qEnd"]), int(got_f[i][k]["dbEnd"])) == (w["score"], w["qEnd"], w["dbEnd"]), (atype, LENGTHS[i], k)
    # a selection writes exactly the selected entries; another set of queries after a forward call must not reuse its images
    sels = [np.array(sorted(rng.choice(len(q[6]), size=(len(q[6]) + 1) // 2, replace=False)), np.int32) for q in queries]
    got = ctx.sw_multi_dir_c(t3, tA if use_aa else None, queries, 1, selections=sels)
    for i in range(len(LENGTHS)):
        mask = np.zeros(len(queries[i][6]), bool); mask[sels[i]] = True
        _same(got[i][mask], want_r[i][mask], (atype, "selected", LENGTHS[i]))
        assert (got[i]["score"][~mask] == 0).all() and (got[i]["word"][~mask] == 0).all()
    shifted = queries[1:] + queries[:1]
    got = ctx.sw_multi_dir_c(t3, tA if use_aa else None, shifted, 1)
    for i in range(len(LENGTHS)):
        _same(got[i], (want_r[1:] + want_r[:1])[i], (atype, "other queries, reversed first", LENGTHS[(i + 1) % len(LENGTHS)]))


def test_sw3_gap_costs_and_int16_saturation(env, shape):
    """other gap costs; position biases large enough to saturate int16: those pairs are re-run in int32 like alignScoreEndPos does"""
    rng, db, ctx, seeds = env
    rng = np.random.default_rng(9)
    mA, m3 = api.Matrix(1, 1.4), api.Matrix(0, 2.1)
    t3, tA = _tiny(m3), _tiny(mA)
    for go, ge in ((10, 1), (8, 2), (15, 3), (3, 1)):
        queries, want = [], []
        for L in (90, 350, 700):
            q3, qa = rng.choice(20, size=L).astype(np.uint8), rng.choice(20, size=L).astype(np.uint8)
            cb = [rng.integers(-3, 4, size=L).astype(np.int8) for _ in range(4)]
            ids = rng.choice(db.n, size=37, replace=False).astype(np.uint32)
            queries.append((qa, q3, cb[0], cb[1], cb[2], cb[3], ids))
            want.append(ctx.sw_batch(_profiles(tA, qa, cb[0]), _profiles(t3, q3, cb[1]), _profiles(tA, qa, cb[2], True), _profiles(t3, q3, cb[3], True), ids, go, ge))
        for d in (0, 1):
            got = ctx.sw_multi_dir_c(t3, tA, queries, d, gap_open=go, gap_extend=ge)
            for i in range(3):
                _same(got[i], want[i][d], (go, ge, d, i))
    # saturation: +100 on every position -> scores beyond 32767 for targets of a few hundred residues
    L = 400
    q3, qa = rng.choice(20, size=L).astype(np.uint8), rng.choice(20, size=L).astype(np.uint8)
    cb = np.full(L, 100, np.int8)
    ids = np.array([db.n - 1, db.n - 2, 0, 5, db.n // 2, db.n - 3], np.uint32)
    want = ctx.sw_batch(_profiles(tA, qa, cb), _profiles(t3, q3, cb), _profiles(tA, qa, cb, True), _profiles(t3, q3, cb, True), ids)
    assert (want[0]["score"] > 32767).any() and (want[0]["score"] < 32767).any()
    for d in (0, 1):
        got = ctx.sw_multi_dir_c(t3, tA, [(qa, q3, cb, cb, cb, cb, ids)], d)
        _same(got[0], want[d], ("saturated", d))
        assert (got[0]["word"] == want[d]["word"]).all()
    both = ctx.sw_multi_c(t3, tA, [(qa, q3, cb, cb, cb, cb, ids)])
    for d in (0, 1):
        _same(both[d][0], want[d], ("saturated, one submission", d))


def test_sw3_large_batch_of_small_hit_lists(env, shape):
    """the all-vs-all shape: hundreds of queries with a handful of pairs each, many classes in one call"""
    rng, db, ctx, seeds = env
    rng = np.random.default_rng(31)
    mA, m3 = api.Matrix(1, 1.4), api.Matrix(0, 2.1)
    t3, tA = _tiny(m3), _tiny(mA)
    qids = rng.choice(db.n, size=300, replace=False)
    queries, want = [], []
    for qi in qids:
        qa, q3 = helpers.target_seqs(db, int(qi))
        if len(q3) == 0:
            continue
        _, _, cba_f, cb3_f = api.align_profiles(mA, m3, qa, q3, True, 0.5)
        _, _, cba_r, cb3_r = api.align_profiles(mA, m3, qa[::-1].copy(), q3[::-1].copy(), True, 0.5)
        ids = np.concatenate([[qi], rng.choice(db.n, size=int(rng.integers(0, 9)), replace=False)]).astype(np.uint32)
        queries.append((qa, q3, cba_f, cb3_f, cba_r, cb3_r, ids))
    got_f = ctx.sw_multi_dir_c(t3, tA, queries, 0)
    got_r = ctx.sw_multi_dir_c(t3, tA, queries, 1)
    both_f, both_r = ctx.sw_multi_c(t3, tA, queries)            # both directions in one submission
    for k in range(len(queries)):
        assert both_f[k].tobytes() == got_f[k].tobytes() and both_r[k].tobytes() == got_r[k].tobytes(), k
    for k in range(0, len(queries), 7):          # every 7th query against the per-pair kernel
        qa, q3, cba_f, cb3_f, cba_r, cb3_r, ids = queries[k]
        f, r = ctx.sw_batch(_profiles(tA, qa, cba_f), _profiles(t3, q3, cb3_f), _profiles(tA, qa, cba_r, True), _profiles(t3, q3, cb3_r, True), ids)
        _same(got_f[k], f, ("fwd", k, len(q3)))
        _same(got_r[k], r, ("rev", k, len(q3)))


@pytest.mark.parametrize("atype", [0, 2])
def test_sw3_call_of_100k_pairs_takes_the_16_lane_shape_by_itself(env, atype, monkeypatch):
    """the automatic rule: a call of >= 100 000 pairs runs its queries of up to 384 rows with 16 lanes per pair (targets of up to 512 columns) -- the records
    must be those of the same call with the shape switched off, and a sample of them the per-pair kernel's"""
    rng, db, ctx, seeds = env
    rng = np.random.default_rng(4100 + atype)
    mA, m3 = api.Matrix(1, 1.4 if atype == 2 else 0.0), api.Matrix(0, 2.1)
    t3, tA = _tiny(m3), _tiny(mA)
    use_aa = atype == 2
    queries = []
    for k in range(104):
        L = int(rng.integers(40, 500))                 # rows-per-lane classes 3 .. 24 of the 16-lane shape, and queries beyond 384 rows in the same call
        q3, qa = rng.choice(20, size=L).astype(np.uint8), rng.choice(20, size=L).astype(np.uint8)
        _, _, cba_f, cb3_f = api.align_profiles(mA, m3, qa, q3, True, 0.5)
        _, _, cba_r, cb3_r = api.align_profiles(mA, m3, qa[::-1].copy(), q3[::-1].copy(), True, 0.5)
        ids = rng.choice(db.n, size=1000, replace=False).astype(np.uint32)
        queries.append((qa if use_aa else None, q3, cba_f if use_aa else None, cb3_f, cba_r if use_aa else None, cb3_r, ids))
    assert sum(len(q[6]) for q in queries) >= 100000
    for k in ("FSGPU_SW3_MID", "FSGPU_SW3_SHORT"):
        monkeypatch.delenv(k, raising=False)
    auto_f = ctx.sw_multi_dir_c(t3, tA if use_aa else None, queries, 0)
    auto_r = ctx.sw_multi_dir_c(t3, tA if use_aa else None, queries, 1)
    monkeypatch.setenv("FSGPU_SW3_MID", "0")
    off_f = ctx.sw_multi_dir_c(t3, tA if use_aa else None, queries, 0)
    off_r = ctx.sw_multi_dir_c(t3, tA if use_aa else None, queries, 1)
    for k in range(len(queries)):
        assert auto_f[k].tobytes() == off_f[k].tobytes() and auto_r[k].tobytes() == off_r[k].tobytes(), k
    for k in (0, 17, 55, 103):
        qa, q3, cba_f, cb3_f, cba_r, cb3_r, ids = queries[k]
        pAf, pAr = (_profiles(tA, qa, cba_f), _profiles(tA, qa, cba_r, True)) if use_aa else (None, None)
        f, r = ctx.sw_batch(pAf, _profiles(t3, q3, cb3_f), pAr, _profiles(t3, q3, cb3_r, True), ids[:64])
        _same(auto_f[k][:64], f, ("fwd", k, len(q3)))
        _same(auto_r[k][:64], r, ("rev", k, len(q3)))
