"""GPU parity of the k-mer prefilter (fsgpu_kmer_index_build / fsgpu_kmer_search through the C ABI) against the
C oracle (oracle/fs_kmer_oracle.c, itself pinned to the compiled reference by test_kmer_oracle_vs_ref.py):
index table, masked lookup, extended 3-mer rows and complete hit lists (ids, scores, diagonals, order), bit-exact,
over the option space that changes the reference's arrival-order rules (BINSIZE, databaseHits refills, result
truncation, 255-capped rescoring, identity hit)."""
import numpy as np
import pytest

import helpers as H
import kmer_lib as K
from foldseek_amd import api, synth

pytestmark = pytest.mark.gpu

N, NQ = 3000, 6


@pytest.fixture(scope="module")
def world():
    O = K.load_ora()
    q3, qa = synth.make_queries(NQ, seed=1)
    db = synth.make_db(N, (q3, qa), homologs_per_query=30, mask_frac=0.02)
    targets = [db.seq(i, "3di", unmask=False) for i in range(db.n)]
    ksub, pb = H.o_submat("MAT3DI", 8.0, -0.2)
    usub, _ = H.o_submat("MAT3DI", 2.0, -0.2)
    o = K.OraKpf(O, ksub, pb, usub, targets)
    ctx = api.Context(0)
    ctx.load_db(db)
    m8 = api.Matrix(0, 8.0, -0.2)
    m2 = api.Matrix(0, 2.0, -0.2)
    assert (m8.scores().ravel() == ksub).all() and (m2.scores().ravel() == usub).all()
    ctx.kmer_index_build(m8, kmer_thr=78)
    yield dict(o=o, ctx=ctx, db=db, q3=q3, m8=m8, m2=m2, targets=targets)
    o.close()


def test_index_matches_oracle(world):
    o, ctx, db = world["o"], world["ctx"], world["db"]
    ooff, oseq, opos = o.index()
    off, seq, pos, msk = ctx.kmer_index_reference_order(db.data3di.size)
    assert ctx.kmer_index_entries == int(ooff[-1])
    assert (off == ooff).all()
    assert (seq == oseq).all() and (pos == opos).all()
    for i in range(0, db.n, 17):
        L = int(db.lengths[i])
        assert (msk[db.offsets[i]:db.offsets[i] + L] == o.masked(i, L)).all(), i


def test_extended_rows(world):
    o, ctx = world["o"], world["ctx"]
    for idx in [0, 1, 19, 20, 399, 400, 7999, 1234, 4321, 6789]:
        s, ix = ctx.kmer_row(idx)
        os_, oi = o.row(3, idx)
        assert (s == os_).all() and (ix.astype(np.uint32) == oi).all(), idx


VARIANTS = [
    dict(),
    dict(maxResListLen=50),
    dict(maxResListLen=50, bins=4),
    dict(maxResListLen=50, bins=16),
    dict(maxResListLen=5),
    dict(maxResListLen=300, maxDbMatches=20000),
    dict(maxResListLen=100, maxDbMatches=9000, bins=8),
    dict(maxResListLen=300, maxDbMatches=5000, foundDiagonalsSize=40000),
    dict(compBias=0, maxResListLen=20, maxDbMatches=15000),
    dict(minDiagScoreThr=15, maxResListLen=2000),
]


def run_gpu(world, kw, queries, ident):
    ctx, m8, m2 = world["ctx"], world["m8"], world["m2"]
    prep = [api.kmer_query_prepare(m8, m2, q, comp_bias=bool(kw.get("compBias", 1)), scale=0.15, kmer_thr=78) for q in queries]
    return ctx.kmer_search(prep, identity=ident, max_res=kw.get("maxResListLen", 1000), min_diag=kw.get("minDiagScoreThr", 30),
                           bins=kw.get("bins", 0), max_db_matches=kw.get("maxDbMatches", 0),
                           found_diagonals_size=kw.get("foundDiagonalsSize", 0), l2_cache_size=2 * 1024 * 1024, want_stats=True)


@pytest.mark.parametrize("kw", VARIANTS)
def test_hit_lists(world, kw):
    o, q3 = world["o"], world["q3"]
    base = dict(maxResListLen=1000, bins=0, maxDbMatches=0, foundDiagonalsSize=0, compBias=1, minDiagScoreThr=30)
    base.update(kw)
    o.set(**base)
    ident = np.array([-1, 7, -1, 100, -1, -1], np.int64)
    orr, os_ = o.run(q3, ident)
    res, status, stats = run_gpu(world, kw, q3, ident)
    for q in range(NQ):
        assert status[q] == 0, (q, status[q])
        assert np.allclose(stats[q], os_[q]), (q, stats[q], os_[q])
        a, b = res[q], orr[q]
        assert len(a) == len(b), (q, kw, len(a), len(b))
        assert (a["id"] == b["id"]).all() and (a["score"] == b["score"]).all() and (a["diag"] == b["diag"]).all(), (q, kw)
    if "maxDbMatches" in kw:
        assert os_[:, 2].sum() > 0


@pytest.mark.parametrize("kw", [dict(), dict(maxResListLen=300, maxDbMatches=5000, foundDiagonalsSize=40000)])
def test_every_bin_level_gives_the_same_hits(world, kw, monkeypatch):
    """the hit-stream partition may cut the targets into coarse keys of any granularity (the residue-balanced default, or 1 / 2 / 4 ... blocks of
    1024 ids per key): the (query, chunk, key) runs change, the hits do not (incl. the databaseHits refill rounds of the second parameter set)"""
    o, q3, ctx = world["o"], world["q3"], world["ctx"]
    base = dict(maxResListLen=1000, bins=0, maxDbMatches=0, foundDiagonalsSize=0, compBias=1, minDiagScoreThr=30)
    base.update(kw)
    o.set(**base)
    qs = list(q3) + [q3[0][:11], q3[1][5:18]]          # two queries of 2 and 4 k-mer positions: a few dozen hits
    orr, _ = o.run(qs, None)
    seen = set()
    for level in (0, 1, 2, 3, 4, 99):
        monkeypatch.setenv("FSGPU_KMER_BIN_LEVEL", str(level))
        res, status, _ = run_gpu(world, kw, qs, None)
        seg = ctx.kmer_segments()
        assert seg[1] == seg[4] and seg[4] % seg[5] == 0 and seg[4] // seg[5] >= len(qs) and 1 <= seg[6] <= 65536, seg
        seen.add(int(seg[5]))
        for q in range(len(qs)):
            assert status[q] == 0 and len(res[q]) == len(orr[q]) and (res[q] == orr[q]).all(), (level, q, seg)
    monkeypatch.delenv("FSGPU_KMER_BIN_LEVEL")
    assert len(seen) >= 2, seen               # more than one granularity occurred


def test_tiny_database_one_key_one_round():
    """60 short targets, short queries: a query's hits number a few dozen -- one coarse key, one tile and one round of k_kmer_dup_stream per query"""
    O = K.load_ora()
    q3, qa = synth.make_queries(5, seed=11, mean_len=30, lo=12, hi=40)
    db = synth.make_db(60, (q3, qa), seed=12, homologs_per_query=3, mean_len=40, lo=12, hi=80)
    targets = [db.seq(i, "3di", unmask=False) for i in range(db.n)]
    ksub, pb = H.o_submat("MAT3DI", 8.0, -0.2)
    usub, _ = H.o_submat("MAT3DI", 2.0, -0.2)
    o = K.OraKpf(O, ksub, pb, usub, targets, kmerThr=100, maxResListLen=20, minDiagScoreThr=15)
    ctx = api.Context(0)
    ctx.load_db(db)
    m8, m2 = api.Matrix(0, 8.0, -0.2), api.Matrix(0, 2.0, -0.2)
    ctx.kmer_index_build(m8, kmer_thr=100)
    prep = [api.kmer_query_prepare(m8, m2, q, kmer_thr=100) for q in q3]
    res, status = ctx.kmer_search(prep, max_res=20, min_diag=15, l2_cache_size=2 << 20)
    seg = ctx.kmer_segments()
    assert seg[5] == 1 and seg[4] == len(q3) and seg[3] <= len(q3) and seg[6] == 60, seg
    want, _ = o.run(q3, None)
    for q in range(len(q3)):
        assert status[q] == 0 and len(res[q]) == len(want[q]) and (res[q] == want[q]).all(), q
    assert sum(len(w) for w in want) > 0
    o.close()
    ctx.close()


def test_large_batches_equal_small_ones(world):
    """more than 32 queries in one device batch (the candidate key has 32 - tbits query bits): 150 queries, many of them repeated with
    another identity id, against the same queries in batches of 7"""
    q3 = world["q3"]
    qs = [q3[i % NQ][: len(q3[i % NQ]) - (i // NQ) % 5] for i in range(150)]
    ident = np.array([(-1 if i % 3 else (i * 7) % N) for i in range(150)], np.int64)
    big, st_big, _ = run_gpu(world, dict(maxResListLen=200), qs, ident)
    assert world["ctx"].kmer_segments()[4] >= 150 * world["ctx"].kmer_segments()[5]              # every query in ONE device batch (runs = (query, chunk) pairs x keys)
    for b in range(0, 150, 7):
        small, st, _ = run_gpu(world, dict(maxResListLen=200), qs[b:b + 7], ident[b:b + 7])
        for k in range(len(small)):
            assert st[k] == st_big[b + k] == 0 and len(small[k]) == len(big[b + k]) and (small[k] == big[b + k]).all(), (b, k)


def test_batching_is_transparent(world):
    """one query per call == all queries in one batch"""
    q3 = world["q3"]
    res_all, st_all, _ = run_gpu(world, {}, q3, None)
    for q in range(NQ):
        r1, s1, _ = run_gpu(world, {}, [q3[q]], None)
        assert (r1[0] == res_all[q]).all()


def test_degenerate_queries(world):
    """shorter than the spaced pattern, all X, and a query with masked (lower-case) residues"""
    o = world["o"]
    o.set(maxResListLen=1000, bins=0, maxDbMatches=0, foundDiagonalsSize=0, compBias=1, minDiagScoreThr=30)
    qs = [np.array([1, 2, 3, 4, 5], np.uint8), np.full(50, 20, np.uint8), world["targets"][40].copy(), np.zeros(0, np.uint8)]
    ident = np.array([-1, 3, 40, -1], np.int64)
    res, status, _ = run_gpu(world, {}, qs, ident)
    for i, q in enumerate(qs):
        if len(q) == 0:
            assert len(res[i]) == 0
            continue
        b, _ = o.query(q, int(ident[i]))
        assert status[i] == 0 and len(res[i]) == len(b) and (res[i] == b).all(), i


def test_low_threshold_long_row_prefixes():
    """a low k-mer threshold makes more than 1024 entries of the sorted 3-mer rows pass: the global-memory instantiation of
    the list kernel (k_kmer_lists<true>) and the k-mer cap (MAX_KMER_RESULT_SIZE) path against the oracle"""
    O = K.load_ora()
    q3, qa = synth.make_queries(3, seed=77, mean_len=60, lo=40, hi=80)
    db = synth.make_db(400, (q3, qa), seed=78, homologs_per_query=10, mean_len=120, lo=30, hi=300)
    targets = [db.seq(i, "3di", unmask=False) for i in range(db.n)]
    ksub, pb = H.o_submat("MAT3DI", 8.0, -0.2)
    usub, _ = H.o_submat("MAT3DI", 2.0, -0.2)
    for thr in (25, 0):
        o = K.OraKpf(O, ksub, pb, usub, targets, kmerThr=thr, maxResListLen=100)
        ctx = api.Context(0)
        ctx.load_db(db)
        m8, m2 = api.Matrix(0, 8.0, -0.2), api.Matrix(0, 2.0, -0.2)
        ctx.kmer_index_build(m8, kmer_thr=thr)
        qs = q3 if thr else [q3[0][:16]]
        prep = [api.kmer_query_prepare(m8, m2, q, kmer_thr=thr) for q in qs]
        res, status, stats = ctx.kmer_search(prep, max_res=100, l2_cache_size=2 << 20, want_stats=True)
        for i, q in enumerate(qs):
            b, st = o.query(q, -1)
            assert status[i] == 0 and np.allclose(stats[i], st), (thr, i, stats[i], st)
            assert len(res[i]) == len(b) and (res[i] == b).all(), (thr, i)
        assert stats[:, 0].max() > 1e5
        o.close()
        ctx.close()


@pytest.mark.parametrize("thr", [130, 100, 78, 25, 0])
def test_wave_and_workgroup_forms_of_the_similar_kmer_passes(thr, monkeypatch):
    """k_kmer_count / k_kmer_lists exist in two forms -- one workgroup per query position, and one WAVE per position with the passing row
    prefixes staged in LDS (the form a low-sensitivity batch takes, e.g. the cluster workflow's -s 1 / -s 4.5 steps) -- picked from the
    number of similar k-mers per position.  Both forced in turn, from thresholds that leave a handful of similar k-mers per position
    to thresholds that make more than 512 row entries pass (the wave form then searches that row in global memory) and reach the
    k-mer cap: same statistics, same hit lists as the oracle."""
    O = K.load_ora()
    q3, qa = synth.make_queries(4, seed=77, mean_len=60, lo=40, hi=80)
    db = synth.make_db(400, (q3, qa), seed=78, homologs_per_query=10, mean_len=120, lo=30, hi=300)
    targets = [db.seq(i, "3di", unmask=False) for i in range(db.n)]
    ksub, pb = H.o_submat("MAT3DI", 8.0, -0.2)
    usub, _ = H.o_submat("MAT3DI", 2.0, -0.2)
    o = K.OraKpf(O, ksub, pb, usub, targets, kmerThr=thr, maxResListLen=100)
    ctx = api.Context(0)
    ctx.load_db(db)
    m8, m2 = api.Matrix(0, 8.0, -0.2), api.Matrix(0, 2.0, -0.2)
    ctx.kmer_index_build(m8, kmer_thr=thr)
    qs = (list(q3) + [q3[0][:7], q3[1][3:9]]) if thr else [q3[0][:16]]
    want = [o.query(q, -1) for q in qs]
    prep = [api.kmer_query_prepare(m8, m2, q, kmer_thr=thr) for q in qs]
    for form in ("0", "1"):
        monkeypatch.setenv("FSGPU_KMER_WAVE", form)
        res, status, stats = ctx.kmer_search(prep, max_res=100, l2_cache_size=2 << 20, want_stats=True)
        for i in range(len(qs)):
            b, st = want[i]
            assert status[i] == 0 and np.allclose(stats[i], st), (thr, form, i, stats[i], st)
            assert len(res[i]) == len(b) and (res[i] == b).all(), (thr, form, i)
    monkeypatch.delenv("FSGPU_KMER_WAVE")
    o.close()
    ctx.close()


def test_unstable_sort_branch_matches_compiled_reference():
    """resultSize >= foundDiagonalsSize/2: the reference leaves its radix path for std::sort (QueryMatcher.cpp:205-215).
    The C oracle does not model libstdc++'s introsort, so this one is checked against the compiled reference itself
    (oracle/_ref travels to the GPU box as a built .so); skipped where it was not built.  A database without planted
    homologs keeps the candidates per target near one, which is what lets the buffer sit between the two limits."""
    R = K.load_ref()
    if R is None:
        pytest.skip("oracle/_ref not built")
    q3, qa = synth.make_queries(4, seed=91)
    db = synth.make_db(4000, None, seed=92, mask_frac=0.0)
    targets = [db.seq(i, "3di", unmask=False) for i in range(db.n)]
    ctx = api.Context(0)
    ctx.load_db(db)
    m8, m2 = api.Matrix(0, 8.0, -0.2), api.Matrix(0, 2.0, -0.2)
    ctx.kmer_index_build(m8, kmer_thr=78)
    w = dict(ctx=ctx, m8=m8, m2=m2)
    # find a foundDiagonalsSize with  candidates < size <= 2 * (targets carrying a diagonal)  for at least one query
    pick = None
    for size in range(8000, 400, -100):
        res, status, stats = run_gpu(w, dict(maxResListLen=60, foundDiagonalsSize=size, bins=2), q3, None)
        if (status >= 0).all() and (status == 1).any():
            pick = (size, res, status)
            break
    assert pick is not None, "no buffer size puts this workload into the std::sort branch"
    size, res, status = pick
    r = K.RefKpf(R, targets, threads=8, maxResListLen=60, foundDiagonalsSize=size, bins=2)
    rr, rs, _ = r.run(q3, None)
    r.close()
    for q in range(len(q3)):
        assert len(res[q]) == len(rr[q]) and (res[q] == rr[q]).all(), (q, size, status[q])
    ctx.close()


def test_masked_query_through_the_rescoring_path(world):
    """a database member carrying soft-mask flags (+32) as query with --max-seqs 1: the cut sits at 255, so rescoreHits
    computes the query's self score from its UNMASKED codes (found by tools/kmer_fuzz.py)"""
    o = world["o"]
    o.set(maxResListLen=1, bins=0, maxDbMatches=3000, foundDiagonalsSize=0, compBias=1, minDiagScoreThr=30)
    cands = [i for i, t in enumerate(world["targets"]) if (t >= 32).any() and len(t) > 200][:4]
    assert cands
    qs = [world["targets"][i].copy() for i in cands]
    res, status, _ = run_gpu(world, dict(maxResListLen=1, maxDbMatches=3000), qs, None)
    for k, q in enumerate(qs):
        b, _ = o.query(q, -1)
        assert status[k] == 0 and len(res[k]) == len(b) == 1 and (res[k] == b).all(), (k, res[k], b)
        assert res[k][0]["id"] == cands[k] and res[k][0]["score"] > 255


def test_low_complexity_database():
    """3Di strings with long runs (sticky Markov chain, like helices / strands): repeat masking, popular k-mers with long
    index lists, many more double-diagonal candidates and natural databaseHits refills at small N"""
    O = K.load_ora()
    rs = np.random.default_rng(5)
    q3, qa = synth.make_queries(4, seed=61)
    q3 = [synth.sticky(rs, q, 0.6) for q in q3]
    db = synth.make_db(2500, (q3, qa), seed=62, homologs_per_query=20, stay=0.6)
    targets = [db.seq(i, "3di", unmask=False) for i in range(db.n)]
    ksub, pb = H.o_submat("MAT3DI", 8.0, -0.2)
    usub, _ = H.o_submat("MAT3DI", 2.0, -0.2)
    o = K.OraKpf(O, ksub, pb, usub, targets, maxResListLen=300, maxDbMatches=60000)
    ctx = api.Context(0)
    ctx.load_db(db)
    m8, m2 = api.Matrix(0, 8.0, -0.2), api.Matrix(0, 2.0, -0.2)
    ctx.kmer_index_build(m8, kmer_thr=78)
    off, seq, pos, msk = ctx.kmer_index_reference_order(db.data3di.size)
    ooff, oseq, opos = o.index()
    assert (off == ooff).all() and (seq == oseq).all() and (pos == opos).all()
    assert (msk == 20).sum() > (db.data3di == 20).sum() + 1000          # repeat masking really happened
    res, status, stats = run_gpu(dict(ctx=ctx, m8=m8, m2=m2), dict(maxResListLen=300, maxDbMatches=60000), q3, None)
    for i, q in enumerate(q3):
        b, st = o.query(q, -1)
        assert status[i] == 0 and np.allclose(stats[i], st)
        assert len(res[i]) == len(b) and (res[i] == b).all(), i
    assert stats[:, 2].sum() > 0
    o.close()
    ctx.close()


def test_no_index_hits_then_next_call():
    """a query without any index hit (single tiny target) must not leave a sticky HIP error behind that the next library call
    would trip over (found by tools/edge_probe.py: unrecorded stage events)"""
    rng = np.random.default_rng(3)
    lens = np.array([50], np.int32)
    db = synth.PaddedDB(np.concatenate([rng.integers(0, 20, 50).astype(np.uint8), np.full(2, 20, np.uint8)]), None,
                        np.array([0, 52], np.int64), lens)
    m8, m2 = api.Matrix(0, 8.0, -0.2), api.Matrix(0, 2.0, -0.2)
    q = rng.integers(0, 20, 120).astype(np.uint8)
    for _ in range(2):
        ctx = api.Context(0)
        ctx.load_db(db)
        ctx.kmer_index_build(m8, kmer_thr=78)
        res, status = ctx.kmer_search([api.kmer_query_prepare(m8, m2, q)], max_res=10)
        assert status[0] == 0 and len(res[0]) == 0
        ctx.close()


@pytest.mark.parametrize("thr,kw", [(78, dict(maxResListLen=100, minDiagScoreThr=0, bins=2)), (78, dict(maxResListLen=7, minDiagScoreThr=0, bins=16)),
                                    (78, dict(maxResListLen=1000, minDiagScoreThr=3, bins=4)), (123, dict(maxResListLen=100, minDiagScoreThr=0, bins=2)),
                                    (154, dict(maxResListLen=100, minDiagScoreThr=0, bins=2))])
def test_kmer_score_only_mode_equals_the_compiled_reference(thr, kw):
    """--diag-score 0 (`kmerScoreOnly`: the first step of easy-cluster's cascaded prefilter runs with -s 1 --diag-score 0 --min-ungapped-score 0
    --max-seqs 100): no ungapped diagonal scores -- a target's score is the number of its double-diagonal k-mer matches (capped at 255), its
    diagonal that of the first one, the cut comes from the histogram of those counts.  Checked against the compiled reference's QueryMatcher
    built with diagonalScoring = false (oracle/_ref, travels to the GPU box): ids, scores, diagonals, order, the truncation at --max-seqs,
    identity hit with score 255; k-mer thresholds of -s 9.5 / 4.5 / 1."""
    R = K.load_ref()
    if R is None:
        pytest.skip("oracle/_ref not built")
    q3, qa = synth.make_queries(5, seed=3)
    db = synth.make_db(2500, (q3, qa), seed=4, homologs_per_query=40, mask_frac=0.02)
    targets = [db.seq(i, "3di", unmask=False) for i in range(db.n)]
    qs = list(q3) + [targets[11].copy()]
    ident = np.array([-1, -1, 5, -1, -1, 11], np.int64)
    ctx = api.Context(0)
    ctx.load_db(db)
    m8, m2 = api.Matrix(0, 8.0, -0.2), api.Matrix(0, 2.0, -0.2)
    ctx.kmer_index_build(m8, kmer_thr=thr)
    prep = [api.kmer_query_prepare(m8, m2, q, comp_bias=True, scale=0.15, kmer_thr=thr) for q in qs]
    res, status = ctx.kmer_search(prep, identity=ident, max_res=kw["maxResListLen"], min_diag=kw["minDiagScoreThr"], bins=kw["bins"], kmer_score_only=True)
    r = K.RefKpf(R, targets, threads=4, kmerThr=thr, noDiagScore=1, compBias=1, **kw)
    rr, _, _ = r.run(qs, ident)
    r.close()
    total = 0
    for q in range(len(qs)):
        assert status[q] == 0, (q, status[q])
        a, b = res[q], rr[q]
        assert len(a) == len(b), (q, thr, kw, len(a), len(b))
        assert (a["id"] == b["id"]).all() and (a["score"] == b["score"]).all() and (a["diag"] == b["diag"]).all(), (q, thr, kw)
        total += len(a)
    assert total > (20 if thr > 100 or kw["maxResListLen"] < 50 else 200) and res[5][0]["id"] == 11 and res[5][0]["score"] == 255
    # refills of databaseHits in this mode: the reference merges the per-refill counts with mergeScoreDuplicates, whose byte array makes targets
    # come out twice and seeds later bins (CacheFriendlyOperations.cpp:150-180) -- replayed on the device (k_kmer_merge_heads).  Entries that share
    # (score, id) are left in whatever order the reference's std::sort puts them: compared in diagonal order
    canon = lambda a: a[np.lexsort((a["diag"], a["id"], -a["score"].astype(np.int64)))]
    refills = dup = 0
    for kw2 in (dict(maxResListLen=50, minDiagScoreThr=0, bins=2, maxDbMatches=3000), dict(maxResListLen=1000, minDiagScoreThr=0, bins=16, maxDbMatches=1500),
                dict(maxResListLen=300, minDiagScoreThr=2, bins=4, maxDbMatches=8000), dict(maxResListLen=7, minDiagScoreThr=0, bins=64, maxDbMatches=2500)):
        res2, status2, stats2 = ctx.kmer_search(prep, identity=ident, max_res=kw2["maxResListLen"], min_diag=kw2["minDiagScoreThr"], bins=kw2["bins"],
                                                max_db_matches=kw2["maxDbMatches"], kmer_score_only=True, want_stats=True)
        r = K.RefKpf(R, targets, threads=4, kmerThr=thr, noDiagScore=1, compBias=1, **kw2)
        rr2, rs2, _ = r.run(qs, ident)
        r.close()
        for q in range(len(qs)):
            assert status2[q] == 0, (q, status2[q], kw2)
            a, b = res2[q], rr2[q]
            assert len(a) == len(b) and (canon(a) == canon(b)).all(), (q, thr, kw2, len(a), len(b))
            assert np.allclose(stats2[q][:3], rs2[q][:3]), (q, stats2[q], rs2[q])
            refills += rs2[q][2] > 0
            dup += len(b) - len(np.unique(b["id"]))
    assert thr > 100 or (refills >= 12 and dup > 0), (refills, dup)
    ctx.close()


@pytest.mark.parametrize("delta,kw", [(-6, dict(maxResListLen=5000, bins=4)), (-8, dict(maxResListLen=5000, bins=2, maxDbMatches=1000)),
                                      (-8, dict(maxResListLen=40, bins=16, maxDbMatches=1500, compBias=0)), (-10, dict(maxResListLen=300, bins=16)),
                                      (-7, dict(maxResListLen=5000, bins=64, maxDbMatches=800)), (-12, dict(maxResListLen=7, bins=8, maxDbMatches=6000)),
                                      (-8, dict(kmerThr=60, maxResListLen=5000, bins=4, maxDbMatches=2500)), (-9, dict(kmerThr=60, maxResListLen=60, bins=32, maxDbMatches=1200))])
def test_cut_zero_with_diagonal_scores_equals_the_compiled_reference(delta, kw):
    """--min-ungapped-score 0 with diagonal scores: when fewer than --max-seqs targets score above 0 the cut is 0 and the reference also returns
    the elements keepMaxScoreElementOnly hands on with score 0 -- after a target's best element every later zero-score element of it, all of them
    for a target whose best is 0 (CacheFriendlyOperations.cpp:112-148, QueryMatcher.cpp:181-200) -- each with its own diagonal, the remaining
    slots filled in the reference's array order.  Two k-mer matches on one diagonal score above 0 under the real matrix, so the case is produced
    the way the checker can follow: the UNGAPPED matrix is lowered by `delta` on all three sides (oracle/ref_kmer_harness.cpp::ref_kpf_shift_ungapped
    on the compiled reference's own object; the device takes the query profile from the caller), the k-mer matrix, index and similar-k-mer
    lists stay as they are.  Settings with databaseHits refills and with cuts inside the zero elements included."""
    R = K.load_ref()
    if R is None or not hasattr(R, "ref_kpf_shift_ungapped"):
        pytest.skip("oracle/_ref not built")
    q3, qa = synth.make_queries(6, seed=5, mean_len=100, lo=10, hi=300)
    db = synth.make_db(1500, (q3, qa), seed=6, homologs_per_query=20, mask_frac=0.1, mean_len=100, lo=8, hi=400)
    targets = [db.seq(i, "3di", unmask=False) for i in range(db.n)]
    ident = np.array([-1, 3, -1, -1, 7, -1], np.int64)
    full = dict(kmerThr=78, minDiagScoreThr=0, compBias=1)
    full.update(kw)
    r = K.RefKpf(R, targets, threads=4, **full)
    R.ref_kpf_shift_ungapped(r.h, delta)
    rr, rs, _ = r.run(list(q3), ident)
    r.close()
    ctx = api.Context(0)
    ctx.load_db(db)
    m8, m2 = api.Matrix(0, 8.0, -0.2), api.Matrix(0, 2.0, -0.2)
    ctx.kmer_index_build(m8, kmer_thr=full["kmerThr"])
    prep = []
    for q in q3:
        s, thr, prof = api.kmer_query_prepare(m8, m2, q, comp_bias=bool(full["compBias"]), scale=0.15, kmer_thr=full["kmerThr"])
        prep.append((s, thr, np.clip(prof.astype(np.int32) + delta, -128, 127).astype(np.int8)))
    res, status, stats = ctx.kmer_search(prep, identity=ident, max_res=full["maxResListLen"], min_diag=0, bins=full["bins"],
                                         max_db_matches=full.get("maxDbMatches", 0), want_stats=True)
    canon = lambda a: a[np.lexsort((a["diag"], a["id"], -a["score"].astype(np.int64)))]
    zeros = 0
    for q in range(len(q3)):
        assert status[q] == 0, (q, status[q])
        a, b = res[q], rr[q]
        assert len(a) == len(b) and (canon(a) == canon(b)).all(), (q, delta, kw, len(a), len(b), canon(a)[-5:], canon(b)[-5:])
        assert np.allclose(stats[q][:3], rs[q][:3])
        zeros += int((b["score"] == 0).sum())
    assert zeros > 0 and ("maxDbMatches" not in kw or rs[:, 2].sum() > 0 or kw["maxDbMatches"] > 5000), (zeros, rs[:, 2])
    ctx.close()


def test_find_duplicates_cut_short_equals_the_compiled_reference():
    """foundDiagonals full (CacheFriendlyOperations.cpp:188-283 `doubleElementCount + elementCount >= outputSize`; upstream issue #1092: more
    than half a bin on one diagonal): the reference's findDuplicates walks its bins in order and RETURNS at the first one whose candidates do not
    fit behind what the earlier bins handed on, so that chunk loses the candidates of that bin and of all later ones -- and the query goes on
    with what fitted (later refills see fewer elements, results can be empty).  Replayed per (query, chunk) from per-bin candidate counts
    (fsgpu_kmer.hip: truncation replay) instead of answered with FSGPU_KMER_E_OUTPUT.  Eight buffer / bin / threshold settings on the compiled
    reference's QueryMatcher with its foundDiagonals shrunk (with and without databaseHits refills, truncation at bin 0 = no results at all,
    queries that also take the unstable-sort branch): ids, scores, diagonals, order, statistics."""
    R = K.load_ref()
    if R is None:
        pytest.skip("oracle/_ref not built")
    q3, qa = synth.make_queries(6, seed=21, mean_len=200, lo=40, hi=600)
    db = synth.make_db(3000, (q3, qa), seed=22, homologs_per_query=60, mask_frac=0.05, mean_len=200, lo=20, hi=800, stay=0.5)
    targets = [db.seq(i, "3di", unmask=False) for i in range(db.n)]
    ident = np.array([-1, 3, -1, -1, 7, -1], np.int64)
    m8, m2 = api.Matrix(0, 8.0, -0.2), api.Matrix(0, 2.0, -0.2)
    ctxs = {}
    cut = refilled = 0
    for kw in [dict(foundDiagonalsSize=150, bins=16, maxDbMatches=3000), dict(foundDiagonalsSize=250, bins=4, maxDbMatches=9000), dict(foundDiagonalsSize=400, bins=2),
               dict(foundDiagonalsSize=200, bins=64, maxDbMatches=20000, maxResListLen=50), dict(foundDiagonalsSize=120, bins=8, maxDbMatches=1500, kmerThr=90),
               dict(foundDiagonalsSize=1500, bins=32, kmerThr=60), dict(foundDiagonalsSize=180, bins=2, maxDbMatches=2500, compBias=0),
               dict(foundDiagonalsSize=2500, bins=8, kmerThr=60, maxDbMatches=30000)]:
        full = dict(kmerThr=78, maxResListLen=1000, minDiagScoreThr=15, compBias=1)
        full.update(kw)
        r = K.RefKpf(R, targets, threads=4, **full)
        rr, rs, _ = r.run(list(q3), ident)
        r.close()
        whole = dict(full, foundDiagonalsSize=0)
        r = K.RefKpf(R, targets, threads=4, **whole)
        rr0, _, _ = r.run(list(q3), ident)
        r.close()
        thr = full["kmerThr"]
        if thr not in ctxs:
            ctxs[thr] = api.Context(0)
            ctxs[thr].load_db(db)
            ctxs[thr].kmer_index_build(m8, kmer_thr=thr)
        prep = [api.kmer_query_prepare(m8, m2, q, comp_bias=bool(full["compBias"]), scale=0.15, kmer_thr=thr) for q in q3]
        res, status, stats = ctxs[thr].kmer_search(prep, identity=ident, max_res=full["maxResListLen"], min_diag=full["minDiagScoreThr"], bins=full["bins"],
                                                   max_db_matches=full.get("maxDbMatches", 0), found_diagonals_size=full["foundDiagonalsSize"], want_stats=True)
        for q in range(len(q3)):
            assert status[q] >= 0, (q, status[q], kw)
            a, b = res[q], rr[q]
            assert len(a) == len(b) and (a == b).all(), (q, kw, status[q], len(a), len(b), a[:3], b[:3])
            assert np.allclose(stats[q][:3], rs[q][:3])
            differs = not (len(b) == len(rr0[q]) and (b == rr0[q]).all())
            cut += differs
            refilled += differs and rs[q][2] > 0
    for c in ctxs.values():
        c.close()
    assert cut >= 30 and refilled >= 10, (cut, refilled)
