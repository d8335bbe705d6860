"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against the oracle on the same seeded inputs.
Integer work: every comparison is exact (bit-identical scores, positions, hit lists)."""
import numpy as np
import pytest

from foldseek_amd import api, synth
import helpers

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def small_db():
    q3, qa = synth.make_queries(8, seed=5, mean_len=250, lo=20, hi=500)
    # force a spread of query lengths over the kernel's register-tile variants
    rng = np.random.default_rng(77)
    for i, L in enumerate((20, 100, 130, 230, 260, 350, 390, 512)):      # SW register classes R = 1,2,3,4,6,8 (two lengths each for 6 and 8)
        q3[i] = rng.choice(20, size=L).astype(np.uint8)
        qa[i] = rng.choice(20, size=L).astype(np.uint8)
    db = synth.make_db(2500, (q3, qa), seed=7, homologs_per_query=40, mask_frac=0.02)
    ctx = api.Context(0)
    ctx.load_db(db)
    yield ctx, db, q3, qa
    ctx.close()


def test_db_bookkeeping(small_db):
    ctx, db, _, _ = small_db
    assert ctx.n == db.n
    assert ctx.residues == db.residues


@pytest.mark.parametrize("qi", range(8))
@pytest.mark.parametrize("comp_bias", [True, False])
def test_gapless_scores_and_hits(small_db, qi, comp_bias):
    ctx, db, q3, _ = small_db
    m = api.Matrix(0, 2.0)
    pssm, cap = api.prefilter_profile(m, q3[qi], comp_bias, 0.15)
    hits = ctx.gapless_scan(pssm, cap, min_score=30, identity=-1, max_res=300)
    got = ctx.gapless_scores().astype(np.int32)
    want = helpers.o_ungapped_scores(q3[qi], db, comp_bias)
    assert (got == want).all(), np.flatnonzero(got != want)[:10]
    sel = helpers.o_prefilter_select(want, 30, -1, 300)
    assert len(hits) == len(sel)
    assert (hits["id"] == sel["key"]).all() and (hits["score"] == sel["score"]).all()


def test_gapless_every_register_count():
    """one query per 16-row register count R = 1..56 (L = 16R - 15 ... 16R; R > 32: the untiled long-query instantiations with
    6-wave workgroups) plus the length extremes of some classes and the first row-tiled length (897)"""
    rng = np.random.default_rng(4242)
    lens = sorted(set([1, 2, 15, 16, 17] + [16 * r - int(rng.integers(0, 16)) for r in range(1, 57)] + [16 * r for r in (5, 11, 23, 31, 32, 33, 36, 37, 48, 49, 56)] +
                      [16 * r + 1 for r in (5, 22, 31, 32, 36, 48, 55, 56)]))
    q3 = [rng.choice(20, size=L).astype(np.uint8) for L in lens]
    qa = [rng.choice(20, size=L).astype(np.uint8) for L in lens]
    db = synth.make_db(600, (q3, qa), seed=19, homologs_per_query=6, mask_frac=0.02)
    ctx = api.Context(0)
    ctx.load_db(db)
    m = api.Matrix(0, 2.0)
    for qi, L in enumerate(lens):
        pssm, cap = api.prefilter_profile(m, q3[qi], True, 0.15)
        hits = ctx.gapless_scan(pssm, cap, min_score=20, max_res=100)
        got = ctx.gapless_scores().astype(np.int32)
        want = helpers.o_ungapped_scores(q3[qi], db, True)
        assert (got == want).all(), (L, np.flatnonzero(got != want)[:10])
        sel = helpers.o_prefilter_select(want, 20, -1, 100)
        assert (hits["id"] == sel["key"]).all() and (hits["score"] == sel["score"]).all(), L
    ctx.close()


def test_gapless_unsorted_database():
    """targets in arbitrary length order (an ASCII DB that never went through makepaddedseqdb): stripes are formed along
    the length order through the slot -> target table, scores still land at the DB's own target ids"""
    rng = np.random.default_rng(99)
    q3, qa = synth.make_queries(3, seed=9, mean_len=200, lo=50, hi=400)
    base = synth.make_db(1500, (q3, qa), seed=10, homologs_per_query=20, hi=1500, mask_frac=0.02)
    perm = rng.permutation(base.n)
    lens = base.lengths[perm].astype(np.int32)
    offsets = np.zeros(base.n + 1, np.int64)
    offsets[1:] = np.cumsum((lens + 3) // 4 * 4)
    d3 = np.full(offsets[-1], 20, np.uint8)
    da = np.full(offsets[-1], 20, np.uint8)
    for new, old in enumerate(perm):
        d3[offsets[new]:offsets[new] + lens[new]] = base.data3di[base.offsets[old]:base.offsets[old] + lens[new]]
        da[offsets[new]:offsets[new] + lens[new]] = base.dataaa[base.offsets[old]:base.offsets[old] + lens[new]]
    db = synth.PaddedDB(d3, da, offsets, lens)
    assert not (np.diff(lens) >= 0).all()
    ctx = api.Context(0)
    ctx.load_db(db)
    m = api.Matrix(0, 2.0)
    for qi in range(3):
        pssm, cap = api.prefilter_profile(m, q3[qi], True, 0.15)
        hits = ctx.gapless_scan(pssm, cap, min_score=30, max_res=150)
        got = ctx.gapless_scores().astype(np.int32)
        want = helpers.o_ungapped_scores(q3[qi], db, True)
        assert (got == want).all(), np.flatnonzero(got != want)[:10]
        assert (got[np.argsort(perm)] == helpers.o_ungapped_scores(q3[qi], base, True)).all()      # same scores as the sorted DB
        sel = helpers.o_prefilter_select(want, 30, -1, 150)
        assert (hits["id"] == sel["key"]).all() and (hits["score"] == sel["score"]).all()
    ctx.close()


def test_gapless_identity_and_truncation(small_db):
    ctx, db, q3, _ = small_db
    m = api.Matrix(0, 2.0)
    pssm, cap = api.prefilter_profile(m, q3[3], True, 0.15)
    want = helpers.o_ungapped_scores(q3[3], db, True)
    low = int(np.argmin(want))                       # an identity hit that would not pass the score filter
    for max_res in (1, 7, 50, 100000):
        hits = ctx.gapless_scan(pssm, cap, min_score=30, identity=low, max_res=max_res)
        sel = helpers.o_prefilter_select(want, 30, low, max_res)
        assert len(hits) == len(sel)
        assert (hits["id"] == sel["key"]).all() and (hits["score"] == sel["score"]).all()
    # a high threshold leaves few or no hits
    hits = ctx.gapless_scan(pssm, cap, min_score=254, identity=-1, max_res=10)
    assert len(hits) == len(helpers.o_prefilter_select(want, 254, -1, 10))


@pytest.mark.parametrize("atype", [2, 0])
@pytest.mark.parametrize("qi", range(8))
def test_sw_score_endpos(small_db, qi, atype):
    ctx, db, q3, qa = small_db
    mAA = api.Matrix(1, 1.4 if atype == 2 else 0.0)
    m3 = api.Matrix(0, 2.1)
    pAf, p3f, _, _ = api.align_profiles(mAA, m3, qa[qi], q3[qi], True, 0.5)
    pAr, p3r, _, _ = api.align_profiles(mAA, m3, qa[qi][::-1].copy(), q3[qi][::-1].copy(), True, 0.5)
    rng = np.random.default_rng(qi)
    ids = np.unique(np.concatenate([rng.integers(0, db.n, size=150), np.arange(db.n - 20, db.n), np.arange(0, 20)])).astype(np.uint32)
    fwd, rev = ctx.sw_batch(pAf if atype == 2 else None, p3f, pAr if atype == 2 else None, p3r, ids)
    L = len(q3[qi])
    for k, t in enumerate(ids):
        ta, t3 = helpers.target_seqs(db, int(t))
        for (pA, p3, got) in ((pAf, p3f, fwd[k]), (pAr, p3r, rev[k])):
            want = helpers.o_sw(pA, p3, L, ta, t3)
            assert (got["score"], got["qEnd"], got["dbEnd"], got["word"]) == (want["score"], want["qEnd"], want["dbEnd"], want["word"]), \
                (qi, int(t), got, want)


def test_sw_long_query_row_tiles():
    """query longer than one 512-row tile: borders travel through HBM between tile launches"""
    rng = np.random.default_rng(123)
    q3 = [rng.choice(20, size=L).astype(np.uint8) for L in (700, 1300)]
    qa = [rng.choice(20, size=L).astype(np.uint8) for L in (700, 1300)]
    db = synth.make_db(300, (q3, qa), seed=11, homologs_per_query=30, hi=1500)
    ctx = api.Context(0)
    ctx.load_db(db)
    mAA, m3 = api.Matrix(1, 1.4), api.Matrix(0, 2.1)
    for qi in range(2):
        pAf, p3f, _, _ = api.align_profiles(mAA, m3, qa[qi], q3[qi], True, 0.5)
        pAr, p3r, _, _ = api.align_profiles(mAA, m3, qa[qi][::-1].copy(), q3[qi][::-1].copy(), True, 0.5)
        ids = np.arange(0, db.n, 3, dtype=np.uint32)
        fwd, rev = ctx.sw_batch(pAf, p3f, pAr, p3r, ids)
        for k, t in enumerate(ids):
            ta, t3 = helpers.target_seqs(db, int(t))
            for (pA, p3, got) in ((pAf, p3f, fwd[k]), (pAr, p3r, rev[k])):
                want = helpers.o_sw(pA, p3, len(q3[qi]), ta, t3)
                assert (got["score"], got["qEnd"], got["dbEnd"], got["word"]) == (want["score"], want["qEnd"], want["dbEnd"], want["word"]), \
                    (qi, int(t), got, want)
    ctx.close()


def test_gapless_long_query_row_tiles():
    """long queries: up to 896 residues in one piece (R = 33..56), beyond that row tiles of at most 512 rows whose diagonals continue
    through border arrays in HBM"""
    rng = np.random.default_rng(321)
    lens = (513, 700, 896, 897, 1024, 1300, 2100)
    q3 = [rng.choice(20, size=L).astype(np.uint8) for L in lens]
    qa = [rng.choice(20, size=L).astype(np.uint8) for L in lens]
    db = synth.make_db(500, (q3, qa), seed=12, homologs_per_query=25, hi=2400, mask_frac=0.02)
    ctx = api.Context(0)
    ctx.load_db(db)
    m = api.Matrix(0, 2.0)
    for qi in range(len(lens)):
        pssm, cap = api.prefilter_profile(m, q3[qi], True, 0.15)
        hits = ctx.gapless_scan(pssm, cap, min_score=30, max_res=200)
        want = helpers.o_ungapped_scores(q3[qi], db, True)
        got = ctx.gapless_scores().astype(np.int32)
        assert (got == want).all(), (lens[qi], np.flatnonzero(got != want)[:10])
        sel = helpers.o_prefilter_select(want, 30, -1, 200)
        assert (hits["id"] == sel["key"]).all() and (hits["score"] == sel["score"]).all()
    ctx.close()


def _manual_db(seqs3, seqsa):
    lens = np.array([len(x) for x in seqs3], dtype=np.int32)
    order = np.argsort(lens, kind="stable")
    lens = lens[order]
    padded = (lens + 3) // 4 * 4
    offsets = np.zeros(len(lens) + 1, np.int64)
    offsets[1:] = np.cumsum(padded)
    d3 = np.full(offsets[-1], 20, np.uint8)
    da = np.full(offsets[-1], 20, np.uint8)
    for new, old in enumerate(order):
        d3[offsets[new]:offsets[new] + lens[new]] = seqs3[old]
        da[offsets[new]:offsets[new] + lens[new]] = seqsa[old]
    return synth.PaddedDB(d3, da, offsets, lens)


def test_sw_int16_saturation_rerun():
    """self-alignment of a long sequence exceeds INT16_MAX -> int32 re-run with segLen = ceil(L/8)"""
    rng = np.random.default_rng(9)
    L = 3000
    # high self-scoring letters (3Di 'M', AA 'W') with 10 % noise, composition bias off so the score really climbs
    q3 = np.where(rng.random(L) < 0.1, rng.choice(20, size=L), 10).astype(np.uint8)
    qa = np.where(rng.random(L) < 0.1, rng.choice(20, size=L), 18).astype(np.uint8)
    seqs3 = [rng.choice(20, size=int(l)).astype(np.uint8) for l in rng.integers(50, 2500, size=14)] + [q3.copy(), q3[100:2900].copy()]
    seqsa = [rng.choice(20, size=len(x)).astype(np.uint8) for x in seqs3[:14]] + [qa.copy(), qa[100:2900].copy()]
    db = _manual_db(seqs3, seqsa)
    ctx = api.Context(0)
    ctx.load_db(db)
    mAA, m3 = api.Matrix(1, 1.4), api.Matrix(0, 2.1)
    pAf, p3f, _, _ = api.align_profiles(mAA, m3, qa, q3, False, 0.5)
    pAr, p3r, _, _ = api.align_profiles(mAA, m3, qa[::-1].copy(), q3[::-1].copy(), False, 0.5)
    ids = np.arange(db.n, dtype=np.uint32)
    fwd, rev = ctx.sw_batch(pAf, p3f, pAr, p3r, ids)
    nsat = 0
    for k, t in enumerate(ids):
        ta, t3 = helpers.target_seqs(db, int(t))
        for (pA, p3, got) in ((pAf, p3f, fwd[k]), (pAr, p3r, rev[k])):
            want = helpers.o_sw(pA, p3, L, ta, t3)
            nsat += int(want["word"] == 2)
            assert (got["score"], got["qEnd"], got["dbEnd"], got["word"]) == (want["score"], want["qEnd"], want["dbEnd"], want["word"]), \
                (int(t), got, want)
    assert nsat >= 1
    ctx.close()


def test_unsupported_gap_costs_are_reported(small_db):
    ctx, db, q3, qa = small_db
    mAA, m3 = api.Matrix(1, 1.4), api.Matrix(0, 2.1)
    pAf, p3f, _, _ = api.align_profiles(mAA, m3, qa[1], q3[1], True, 0.5)
    with pytest.raises(api.FsgpuError):
        ctx.sw_batch(pAf, p3f, pAf, p3f, np.arange(4, dtype=np.uint32), gap_open=1, gap_extend=1)


def test_ragged_and_tiny_database():
    """n not a multiple of 8, length-1 targets, a target that is all X / all masked"""
    rng = np.random.default_rng(4)
    lens = np.array([1, 1, 2, 3, 5, 8, 13, 21, 34, 55, 89], dtype=np.int32)
    padded = (lens + 3) // 4 * 4
    offsets = np.zeros(len(lens) + 1, np.int64)
    offsets[1:] = np.cumsum(padded)
    d3 = np.full(offsets[-1], 20, np.uint8)
    da = np.full(offsets[-1], 20, np.uint8)
    for i, l in enumerate(lens):
        d3[offsets[i]:offsets[i] + l] = rng.integers(0, 20, size=l)
        da[offsets[i]:offsets[i] + l] = rng.integers(0, 20, size=l)
    d3[offsets[5]:offsets[5] + 8] = 20                      # all X
    d3[offsets[6]:offsets[6] + 13] += 32                    # all soft-masked
    db = synth.PaddedDB(d3, da, offsets, lens)
    ctx = api.Context(0)
    ctx.load_db(db)
    q3 = rng.integers(0, 20, size=45).astype(np.uint8)
    qa = rng.integers(0, 20, size=45).astype(np.uint8)
    m = api.Matrix(0, 2.0)
    pssm, cap = api.prefilter_profile(m, q3, True, 0.15)
    ctx.gapless_scan(pssm, cap, min_score=0, max_res=100)
    assert (ctx.gapless_scores().astype(np.int32) == helpers.o_ungapped_scores(q3, db, True)).all()
    mAA, m3 = api.Matrix(1, 1.4), api.Matrix(0, 2.1)
    pAf, p3f, _, _ = api.align_profiles(mAA, m3, qa, q3, True, 0.5)
    pAr, p3r, _, _ = api.align_profiles(mAA, m3, qa[::-1].copy(), q3[::-1].copy(), True, 0.5)
    ids = np.arange(db.n, dtype=np.uint32)
    fwd, rev = ctx.sw_batch(pAf, p3f, pAr, p3r, ids)
    for k in range(db.n):
        ta, t3 = helpers.target_seqs(db, k)
        want = helpers.o_sw(pAf, p3f, 45, ta, t3)
        assert (fwd[k]["score"], fwd[k]["qEnd"], fwd[k]["dbEnd"]) == (want["score"], want["qEnd"], want["dbEnd"])
        want = helpers.o_sw(pAr, p3r, 45, ta, t3)
        assert (rev[k]["score"], rev[k]["qEnd"], rev[k]["dbEnd"]) == (want["score"], want["qEnd"], want["dbEnd"])
    # empty hit list
    fwd, rev = ctx.sw_batch(pAf, p3f, pAr, p3r, np.zeros(0, np.uint32))
    assert len(fwd) == 0
    ctx.close()


def test_align_batch_equals_per_query_align():
    """fshost_search_align_batch / fsgpu_sw_multi (one launch per register class for many queries) returns exactly what the
    per-query calls return: mixed lengths across all register classes, a row-tiled long query, an empty hit list, a pair
    that saturates int16 (int32 re-run), both alignment types; complete result records and backtraces."""
    rng = np.random.default_rng(123)
    lens = [40, 70, 130, 200, 260, 300, 350, 380, 420, 447, 500, 512, 700, 33]
    q3 = [rng.choice(20, size=L, p=synth.BACK_3DI / synth.BACK_3DI.sum()).astype(np.uint8) for L in lens]
    qa = [rng.choice(20, size=L, p=synth.BACK_AA / synth.BACK_AA.sum()).astype(np.uint8) for L in lens]
    # a self-scoring monster: 3Di 'M' / AA 'W' repeated -> int16 saturation against its planted copy
    q3.append(np.full(3000, 10, np.uint8)); qa.append(np.full(3000, 18, np.uint8))
    db = synth.make_db(1500, (q3, qa), seed=321, homologs_per_query=12, mask_frac=0.01, hi=3200)
    ctx = api.Context(0)
    ctx.load_db(db)
    for atype in (2, 0):
        par = api.default_params()
        par.alignmentType = atype
        par.addBacktrace = 1
        s = api.Search(ctx, par)
        hit_lists = []
        for i in range(len(q3)):
            ids = rng.choice(db.n, size=int(rng.integers(60, 400)), replace=False).astype(np.uint32)
            hit_lists.append(ids)
        hit_lists[3] = np.zeros(0, np.uint32)
        single = [s.align(qa[i], q3[i], hit_lists[i], with_backtrace=True) for i in range(len(q3))]
        batch, bts = s.align_batch(qa, q3, hit_lists, with_backtrace=True)
        total = 0
        for i in range(len(q3)):
            r1, b1 = single[i]
            assert len(batch[i]) == len(r1), (atype, i)
            for f in ("dbKey", "score", "qcov", "dbcov", "seqId", "eval", "alnLength", "qStartPos", "qEndPos", "qLen", "dbStartPos", "dbEndPos", "dbLen"):
                assert (batch[i][f] == r1[f]).all(), (atype, i, f)
            assert bts[i] == b1, (atype, i)
            total += len(r1)
        assert total > 20
        s.close()
    ctx.close()


def test_sw_multi_dir_two_targets_per_wave():
    """fsgpu_sw_multi_dir (k_sw2: the same direction of two targets in one wave) against fsgpu_sw_batch (k_sw: both
    directions of one target) and the oracle: every register class, odd pair counts, targets of very different length in
    one wave, a one-pair list, with and without AA; a selection writes exactly the selected entries."""
    rng = np.random.default_rng(2024)
    lens = [20, 100, 130, 230, 350, 390, 512]
    q3 = [rng.choice(20, size=L).astype(np.uint8) for L in lens]
    qa = [rng.choice(20, size=L).astype(np.uint8) for L in lens]
    db = synth.make_db(900, (q3, qa), seed=55, homologs_per_query=10, lo=1, hi=1800, mask_frac=0.02)
    ctx = api.Context(0)
    ctx.load_db(db)
    for atype in (0, 2):
        mA, m3 = api.Matrix(1, 1.4 if atype == 2 else 0.0), api.Matrix(0, 2.1)
        queries, want_f, want_r = [], [], []
        for i, L in enumerate(lens):
            pAf, p3f, _, _ = api.align_profiles(mA, m3, qa[i], q3[i], True, 0.5)
            pAr, p3r, _, _ = api.align_profiles(mA, m3, qa[i][::-1].copy(), q3[i][::-1].copy(), True, 0.5)
            n = [1, 7, 64, 33, 120, 9, 250][i]
            ids = rng.choice(db.n, size=n, replace=False).astype(np.uint32)
            if i == 4:
                ids[:4] = [0, db.n - 1, 1, db.n - 2]          # shortest and longest targets of the DB side by side
            use_aa = atype == 2
            queries.append((pAf if use_aa else None, p3f, pAr if use_aa else None, p3r, L, ids))
            f, r = ctx.sw_batch(pAf if use_aa else None, p3f, pAr if use_aa else None, p3r, ids)
            want_f.append(f); want_r.append(r)
        for direction, want in ((0, want_f), (1, want_r)):
            got = ctx.sw_multi_dir(queries, direction)
            for i in range(len(lens)):
                for fld in ("score", "qEnd", "dbEnd"):
                    assert (got[i][fld] == want[i][fld]).all(), (atype, direction, lens[i], fld)
        # oracle spot check of the forward pass
        for i in (1, 4):
            pA, p3 = helpers.o_align_profiles(qa[i], q3[i], atype)[:2]
            for k in range(min(6, len(queries[i][5]))):
                ta, t3 = helpers.target_seqs(db, int(queries[i][5][k]))
                w = helpers.o_sw(pA, p3, lens[i], ta, t3)
                assert (int(want_f[i][k]["score"]), int(want_f[i][k]["qEnd"]), int(want_f[i][k]["dbEnd"])) == (w["score"], w["qEnd"], w["dbEnd"])
        # selections
        sels = [np.array(sorted(rng.choice(len(q[5]), size=len(q[5]) // 3, replace=False)), np.int32) for q in queries]
        got = ctx.sw_multi_dir(queries, 1, selections=sels)
        for i in range(len(lens)):
            mask = np.zeros(len(queries[i][5]), bool); mask[sels[i]] = True
            assert (got[i]["score"][mask] == want_r[i]["score"][mask]).all() and (got[i]["dbEnd"][mask] == want_r[i]["dbEnd"][mask]).all()
            assert (got[i]["score"][~mask] == 0).all() and (got[i]["word"][~mask] == 0).all()
    ctx.close()


def test_sw_multi_dir_row_tiled_queries():
    """Queries longer than 512 rows inside a multi-query call: one k_sw launch per tile level over all of them (SwTileBlock) against
    the single-query path (fsgpu_sw_batch) and the oracle; the reversed call is answered from the forward call's launches, must not be
    when the profiles changed in between, and works on its own; selections; short queries mixed in; int16-saturated pairs re-run."""
    rng = np.random.default_rng(77)
    lens = [513, 600, 1024, 350, 1025, 1600, 3000, 64]
    q3 = [rng.choice(20, size=L).astype(np.uint8) for L in lens]
    qa = [rng.choice(20, size=L).astype(np.uint8) for L in lens]
    db = synth.make_db(700, (q3, qa), seed=56, homologs_per_query=6, lo=1, hi=2500, mask_frac=0.02)
    ctx = api.Context(0)
    ctx.load_db(db)
    npairs = [1, 9, 17, 30, 8, 40, 5, 12]
    for atype in (0, 2):
        mA, m3 = api.Matrix(1, 1.4 if atype == 2 else 0.0), api.Matrix(0, 2.1)
        use_aa = atype == 2

        def build(q3s, qas):
            out = []
            for i, L in enumerate(lens):
                pAf, p3f, _, _ = api.align_profiles(mA, m3, qas[i], q3s[i], True, 0.5)
                pAr, p3r, _, _ = api.align_profiles(mA, m3, qas[i][::-1].copy(), q3s[i][::-1].copy(), True, 0.5)
                out.append([pAf if use_aa else None, p3f, pAr if use_aa else None, p3r, L, None])
            return out
        queries = build(q3, qa)
        hom = [np.nonzero(helpers.o_ungapped_scores(q3[i], db, False) > 200)[0] for i in range(len(lens))]
        want_f, want_r = [], []
        for i in range(len(lens)):
            ids = rng.choice(db.n, size=npairs[i], replace=False).astype(np.uint32)
            k = min(len(hom[i]), max(1, npairs[i] // 3))
            ids[:k] = hom[i][:k]                                     # planted relatives
            ids = np.unique(ids).astype(np.uint32)
            queries[i][5] = ids
            f, r = ctx.sw_batch(*queries[i][:4], ids)
            want_f.append(f); want_r.append(r)

        def check(got, want, what):
            for i in range(len(lens)):
                for fld in ("score", "qEnd", "dbEnd"):
                    assert (got[i][fld] == want[i][fld]).all(), (atype, what, lens[i], fld)
        qs = [tuple(q) for q in queries]
        check(ctx.sw_multi_dir(qs, 0), want_f, "forward")
        check(ctx.sw_multi_dir(qs, 1), want_r, "reversed, from the forward call's launches")
        check(ctx.sw_multi_dir(qs, 1), want_r, "reversed on its own")
        # forward call, then a reversed call whose reversed profiles differ: must be computed, not taken from the forward call
        other = build([x[::-1].copy() for x in q3], [x[::-1].copy() for x in qa])
        mixed = [(q[0], q[1], o[2], o[3], q[4], q[5]) for q, o in zip(queries, other)]
        want_mixed = [ctx.sw_batch(*m[:4], m[5])[1] for m in mixed]
        ctx.sw_multi_dir(qs, 0)
        check(ctx.sw_multi_dir(mixed, 1), want_mixed, "reversed with other profiles after a forward call")
        # selections after a forward call, and without one
        sels = [np.array(sorted(rng.choice(len(q[5]), size=(len(q[5]) + 1) // 2, replace=False)), np.int32) for q in qs]
        for with_forward in (True, False):
            if with_forward:
                ctx.sw_multi_dir(qs, 0)
            got = ctx.sw_multi_dir(qs, 1, selections=sels)
            for i in range(len(lens)):
                mask = np.zeros(len(qs[i][5]), bool); mask[sels[i]] = True
                assert (got[i]["score"][mask] == want_r[i]["score"][mask]).all() and (got[i]["dbEnd"][mask] == want_r[i]["dbEnd"][mask]).all() and (got[i]["qEnd"][mask] == want_r[i]["qEnd"][mask]).all()
                assert (got[i]["score"][~mask] == 0).all() and (got[i]["word"][~mask] == 0).all()
        # oracle spot check (forward, one long query with three tiles and one with two)
        for i in (4, 0):
            pA, p3 = helpers.o_align_profiles(qa[i], q3[i], atype)[:2]
            for k in range(min(4, len(qs[i][5]))):
                ta, t3 = helpers.target_seqs(db, int(qs[i][5][k]))
                if len(t3) == 0:
                    continue
                w = helpers.o_sw(pA, p3, lens[i], ta, t3)
                assert (int(want_f[i][k]["score"]), int(want_f[i][k]["qEnd"]), int(want_f[i][k]["dbEnd"])) == (w["score"], w["qEnd"], w["dbEnd"])
    ctx.close()
    # int16-saturated pairs of a row-tiled query inside a multi-query call: re-run by the int32 kernel, both directions
    rng = np.random.default_rng(9)
    L = 3000
    q3s = np.where(rng.random(L) < 0.1, rng.choice(20, size=L), 10).astype(np.uint8)
    qas = np.where(rng.random(L) < 0.1, rng.choice(20, size=L), 18).astype(np.uint8)
    seqs3 = [rng.choice(20, size=int(l)).astype(np.uint8) for l in rng.integers(50, 2500, size=14)] + [q3s.copy(), q3s[100:2900].copy()]
    seqsa = [rng.choice(20, size=len(x)).astype(np.uint8) for x in seqs3[:14]] + [qas.copy(), qas[100:2900].copy()]
    db = _manual_db(seqs3, seqsa)
    ctx = api.Context(0)
    ctx.load_db(db)
    mA, m3 = api.Matrix(1, 1.4), api.Matrix(0, 2.1)
    qs = []
    for a3, aa in ((q3s, qas), (seqs3[3], seqsa[3]), (q3s[:700].copy(), qas[:700].copy())):
        pAf, p3f, _, _ = api.align_profiles(mA, m3, aa, a3, False, 0.5)
        pAr, p3r, _, _ = api.align_profiles(mA, m3, aa[::-1].copy(), a3[::-1].copy(), False, 0.5)
        qs.append((pAf, p3f, pAr, p3r, len(a3), np.arange(db.n, dtype=np.uint32)))
    want = [ctx.sw_batch(*q[:4], q[5]) for q in qs]
    assert (want[0][0]["word"] == 2).any() and (want[0][1]["word"] == 2).any()
    for direction in (0, 1):
        got = ctx.sw_multi_dir(qs, direction)
        for i in range(len(qs)):
            for fld in ("score", "qEnd", "dbEnd", "word"):
                assert (got[i][fld] == want[i][direction][fld]).all(), (direction, i, fld)
    ctx.close()
