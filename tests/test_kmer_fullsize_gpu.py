"""k-mer prefilter at the size of BASELINE.json configs[1] (100k targets, 35 M residues): size-independent properties of
the GPU hit lists, and equality with the compiled reference's QueryMatcher where oracle/_ref was built (it travels to
the GPU box as a .so).  Some of the longer queries refill databaseHits naturally (> 2 M index hits)."""
import numpy as np
import pytest

import kmer_lib as K
from foldseek_amd import api, synth

pytestmark = pytest.mark.gpu
N, NQ = 100000, 12


@pytest.fixture(scope="module")
def world():
    q3, qa = synth.make_queries(NQ, seed=1)
    q3[5] = np.concatenate([q3[5], q3[6], q3[7]])[:1100]          # a long query: > 2 M index hits -> refills
    qa[5] = np.concatenate([qa[5], qa[6], qa[7]])[:1100]
    db = synth.make_db(N, (q3, qa))
    ctx = api.Context(0)
    ctx.load_db(db)
    m8, m2 = api.Matrix(0, 8.0, -0.2), api.Matrix(0, 2.0, -0.2)
    ctx.kmer_index_build(m8, kmer_thr=78)
    prep = [api.kmer_query_prepare(m8, m2, q) for q in q3]
    R = K.load_ref()
    l2 = int(R.ref_l2_cache_size()) if R is not None else 2 << 20
    ident = np.full(NQ, -1, np.int64)
    ident[2] = 4242
    res, status, stats = ctx.kmer_search(prep, identity=ident, max_res=1000, l2_cache_size=l2, want_stats=True)
    return dict(db=db, q3=q3, ctx=ctx, prep=prep, res=res, status=status, stats=stats, ident=ident, R=R, l2=l2)


def _diag_score(profile, masked_target, diag16):
    """UngappedAlignment::computeSingelSequenceScores restated in numpy"""
    qL, tL = len(profile), len(masked_target)
    d = int(np.int16(np.uint16(diag16)))
    if d >= 0:
        if d >= qL:
            return 0
        n = min(tL, qL - d)
        v = profile[np.arange(d, d + n), masked_target[:n]]
    else:
        if -d >= tL:
            return 0
        n = min(tL + d, qL)
        v = profile[np.arange(n), masked_target[-d:-d + n]]
    best = run = 0
    for x in v.astype(np.int64):
        run = max(0, run + x)
        best = max(best, run)
    return best


def test_properties_at_full_size(world):
    db, res, status, stats = world["db"], world["res"], world["status"], world["stats"]
    assert (status >= 0).all()
    assert stats[:, 2].sum() >= 1                    # at least one query refilled databaseHits
    _, _, masked = world["ctx"].kmer_index_copy(db.data3di.size)[0:3]
    for q in range(NQ):
        h = res[q]
        assert 0 < len(h) <= 1000
        body = h[1:] if world["ident"][q] >= 0 else h
        if world["ident"][q] >= 0:
            assert h[0]["id"] == world["ident"][q] and h[0]["score"] == 65535 and h[0]["diag"] == 0
            assert (body["id"] != world["ident"][q]).all()
        assert len(np.unique(h["id"])) == len(h)
        key = np.abs(body["score"]).astype(np.int64) * (1 << 32) - body["id"].astype(np.int64)
        assert (np.diff(key) < 0).all()                                  # (score desc, id asc), strictly
        assert (body["score"] >= 30).all()
        prof = world["prep"][q][2].astype(np.int64)
        for k in list(range(min(12, len(body)))) + list(range(len(body) - 6, len(body))):
            t = int(body["id"][k])
            mt = masked[db.offsets[t]:db.offsets[t] + db.lengths[t]]
            s = _diag_score(prof, mt, int(body["diag"][k]))
            assert body["score"][k] == s, (q, k, t, int(body["score"][k]), s)


def test_query_order_and_batch_split_do_not_matter(world):
    ctx, prep = world["ctx"], world["prep"]
    order = [7, 3, 11, 0]
    r2, st2 = ctx.kmer_search([prep[i] for i in order], identity=world["ident"][order], max_res=1000, l2_cache_size=world["l2"])
    for k, i in enumerate(order):
        assert (r2[k] == world["res"][i]).all()


def test_equals_compiled_reference_at_full_size(world):
    R = world["R"]
    if R is None:
        pytest.skip("oracle/_ref not built")
    db = world["db"]
    r = K.RefKpf(R, [db.seq(i, "3di", unmask=False) for i in range(db.n)], threads=64)
    rr, rs, _ = r.run(world["q3"], world["ident"], threads=16)
    r.close()
    for q in range(NQ):
        assert len(world["res"][q]) == len(rr[q]) and (world["res"][q] == rr[q]).all(), q
        assert np.allclose(world["stats"][q][:3], rs[q][:3])


def test_equals_compiled_reference_at_1M_targets():
    """configs[2] size: 1M targets (350 M residues, every query refills databaseHits several times).  Six queries against the
    compiled reference's QueryMatcher over the same 1M sequences (index build ~10 s on the box's 16 cores)."""
    R = K.load_ref()
    if R is None:
        pytest.skip("oracle/_ref not built")
    q3, qa = synth.make_queries(6, seed=11, lo=200, hi=420)
    db = synth.make_db_fast(1000000, (q3, qa), seed=271828, homologs_per_query=50)
    ctx = api.Context(0)
    ctx.load_db(db)
    m8, m2 = api.Matrix(0, 8.0, -0.2), api.Matrix(0, 2.0, -0.2)
    ctx.kmer_index_build(m8, kmer_thr=78)
    prep = [api.kmer_query_prepare(m8, m2, q) for q in q3]
    l2 = int(R.ref_l2_cache_size())
    ident = np.full(len(q3), -1, np.int64)
    ident[1] = 999999
    res, status, stats = ctx.kmer_search(prep, identity=ident, max_res=1000, l2_cache_size=l2, want_stats=True)
    assert (status >= 0).all() and stats[:, 2].sum() >= len(q3)          # every query refills at this size
    # targets as views into the padded buffer: no per-entry copies of 350 MB
    r = K.RefKpf(R, [db.data3di[db.offsets[i]:db.offsets[i] + db.lengths[i]] for i in range(db.n)], threads=16)
    rr, rs, _ = r.run(q3, ident, threads=16)
    r.close()
    ctx.close()
    for q in range(len(q3)):
        assert len(res[q]) == len(rr[q]) and (res[q] == rr[q]).all(), q
        assert np.allclose(stats[q][:3], rs[q][:3])


def test_equals_compiled_reference_beyond_16M_targets():
    """More targets than the hit-stream partition took until round 5 (16.4 M): 17 M
    short structures (30 .. 64 residues, 0.6 G residues; 25 id bits, so a device batch holds 128 queries and the index entries stay 4 bytes), four
    queries with planted homologs against the compiled reference's QueryMatcher over the same sequences.  The limit that remains is 33.5 M targets
    (512 coarse keys of at most 65536 ids) and 2^32 residues (k = 6), include/fsgpu.h."""
    R = K.load_ref()
    if R is None:
        pytest.skip("oracle/_ref not built")
    q3, qa = synth.make_queries(4, seed=17, mean_len=52, lo=44, hi=64)
    db = synth.make_db_fast(17000000, (q3, qa), seed=1717, homologs_per_query=40, mean_len=36.0, lo=30, hi=64)
    assert db.n > 16400000
    ctx = api.Context(0)
    ctx.load_db(db)
    m8, m2 = api.Matrix(0, 8.0, -0.2), api.Matrix(0, 2.0, -0.2)
    ctx.kmer_index_build(m8, kmer_thr=78)
    prep = [api.kmer_query_prepare(m8, m2, q) for q in q3]
    l2 = int(R.ref_l2_cache_size())
    ident = np.full(len(q3), -1, np.int64)
    ident[2] = 16999999
    res, status, stats = ctx.kmer_search(prep, identity=ident, max_res=1000, l2_cache_size=l2, want_stats=True)
    assert (status >= 0).all()
    assert ctx.kmer_segments()[5] >= 17000000 // 65536                     # coarse keys hold at most 64 blocks of 1024 ids
    r = K.RefKpf.from_padded(R, db, threads=16)
    rr, rs, _ = r.run(q3, ident, threads=4)
    r.close()
    ctx.close()
    for q in range(len(q3)):
        assert len(res[q]) == len(rr[q]) and (res[q] == rr[q]).all(), q
        assert len(res[q]) >= 20                                           # the planted homologs are found
        assert np.allclose(stats[q][:3], rs[q][:3])
