"""Gapless prefilter + structure SW at the size of BASELINE.json configs[1] (100k targets, 35 M residues) and -- scores
only -- configs[2] (1M targets): the device path against the COMPILED REFERENCE (oracle/_ref/libfsref.so travels to the
GPU box as a built file) and, where the reference would take too long, against size-independent properties.

What only shows up at this size: the column-segment work items and the atomic work queue of the scan (k_gapless.hpp), the
byte-wise atomicCAS max over a stripe's segments, 8-target stripes of up to 2000 columns, multi-query SW launches with
tens of thousands of waves, several register classes in one launch.
"""
import os
import numpy as np
import pytest

import helpers
import oracle_lib
from foldseek_amd import api, synth

pytestmark = pytest.mark.gpu
N = 100000


def _queries():
    q3, qa = synth.make_queries(6, seed=77, lo=120, hi=480)
    q3[3] = np.concatenate([q3[3], q3[4], q3[5]])[:777]            # row-tiled query (> 512 rows)
    qa[3] = np.concatenate([qa[3], qa[4], qa[5]])[:777]
    m3, ma = synth.make_queries(4, seed=78, lo=300, hi=480)
    q3[4], qa[4] = np.concatenate(m3)[:1300], np.concatenate(ma)[:1300]      # three row tiles; two row-tiled queries share the SW tile launches
    q3, qa = q3[:5], qa[:5]
    # round 5: one query per k_sw3<*, 64, *> launch group (513 .. 1024 residues run with 64 lanes per target pair: R = ceil(L / 64) = 10, 15, 16 rows per
    # lane -- the 9..12 and 13..16 LDS classes of the RLO = 9 kernel; 777 above is R = 13), so that each meets ref_structure_align directly
    n3, na = synth.make_queries(9, seed=79, lo=300, hi=480)
    for L, k in ((600, 0), (900, 3), (1024, 6)):
        q3.append(np.concatenate(n3[k:k + 3])[:L]); qa.append(np.concatenate(na[k:k + 3])[:L])
    return q3, qa


@pytest.fixture(scope="module")
def world():
    q3, qa = _queries()
    db = synth.make_db_fast(N, (q3, qa), seed=4711, homologs_per_query=50)
    ctx = api.Context(0)
    ctx.load_db(db)
    ref = oracle_lib.load_ref()
    return dict(db=db, q3=q3, qa=qa, ctx=ctx, ref=ref)


def _ref_scores(ref, db, q, threads=16):
    scores = np.zeros(db.n, np.int32)
    ref.ref_ungapped(q, len(q), 1, 0.15, db.data3di, np.ascontiguousarray(db.offsets[:-1]), np.ascontiguousarray(db.lengths), db.n, threads, scores)
    return scores


def test_gapless_scores_and_hit_lists_equal_reference_at_100k(world):
    """every one of the 100k per-target scores + the selected hit list, eight query lengths (R classes 8..56 and three row-tiled ones)"""
    ref, db = world["ref"], world["db"]
    if ref is None:
        pytest.skip("oracle/_ref not built")
    s = api.Search(world["ctx"])
    for qi, q in enumerate(world["q3"]):
        hits = s.prefilter(q, identity=(12345 if qi == 1 else -1))
        got = world["ctx"].gapless_scores().astype(np.int32)
        want = _ref_scores(ref, db, q)
        bad = np.flatnonzero(got != want)
        assert len(bad) == 0, (qi, len(q), bad[:10], got[bad[:10]], want[bad[:10]])
        sel = helpers.o_prefilter_select(want, 30, 12345 if qi == 1 else -1, 1000)
        assert len(hits) == len(sel) == 1000
        assert (hits["id"] == sel["key"]).all() and (hits["score"] == sel["score"]).all()
    s.close()


def _align_against_reference(world, atype, go, ge, queries, max_hits, device_backtrace=None):
    """device_backtrace: None = the library's own choice, 0 / 1 = host / device block aligner (FSGPU_DEVICE_BACKTRACE): with 1 the CIGARs the
    compiled reference's alignStructure wrote are met by k_block_backtrace directly, not through the host restatement"""
    if device_backtrace is not None:
        os.environ["FSGPU_DEVICE_BACKTRACE"] = str(device_backtrace)
        os.environ["FSGPU_BT_PASS2"] = "1"          # the device's second pass (blocks of up to 512 rows) runs however few hits reach it
    try:
        return _align_against_reference_body(world, atype, go, ge, queries, max_hits, device_backtrace)
    finally:
        os.environ.pop("FSGPU_DEVICE_BACKTRACE", None)
        os.environ.pop("FSGPU_BT_PASS2", None)


def _align_against_reference_body(world, atype, go, ge, queries, max_hits, device_backtrace):
    ref, db, q3, qa = world["ref"], world["db"], world["q3"], world["qa"]
    par = api.default_params()
    par.alignmentType = atype
    par.addBacktrace = 1
    par.gapOpen, par.gapExtend = go, ge
    s = api.Search(world["ctx"], par)
    pre = api.Search(world["ctx"])
    hit_lists = {qi: pre.prefilter(q3[qi])["id"][:max_hits] for qi in queries}
    pre.close()
    res, bts = s.align_batch([qa[qi] for qi in queries], [q3[qi] for qi in queries], [hit_lists[qi] for qi in queries], with_backtrace=True)
    t3 = np.where(db.data3di >= 32, db.data3di - 32, db.data3di).astype(np.uint8)
    accepted = 0
    for slot, qi in enumerate(queries):
        h = np.ascontiguousarray(hit_lists[qi].astype(np.int64))
        n = len(h)
        aln = np.zeros(n, oracle_lib.REFALN_DT)
        cig = np.zeros(1 << 22, np.uint8)
        ref.ref_structure_align(qa[qi], q3[qi], len(q3[qi]), atype, 1, 0.5, go, ge, db.dataaa, t3,
                                np.ascontiguousarray(db.offsets[:-1][h]), np.ascontiguousarray(db.lengths[h]), n,
                                db.residues, 10.0, 1, 16, None, None, aln.ctypes.data, cig.ctypes.data, cig.size)
        cigs = cig.tobytes().split(b"\0")[0].decode().split("\n")
        ok = np.flatnonzero(aln["status"] == 0)
        # reference records in structurealign's output order (Matcher::compareHits: e-value asc, score desc, dbLen asc, key asc)
        want = sorted(ok, key=lambda k: (aln["evalue"][k], -int(aln["score"][k]), int(db.lengths[h[k]]), int(h[k])))
        got = res[slot]
        assert len(got) == len(want), (qi, go, ge, len(got), len(want))
        for r, k, bt in zip(got, want, bts[slot]):
            a = aln[k]
            assert (r["dbKey"], r["score"], r["qStartPos"], r["qEndPos"], r["dbStartPos"], r["dbEndPos"]) == \
                   (h[k], a["score"], a["qStart"], a["qEnd"], a["dbStart"], a["dbEnd"]), (qi, k, go, ge)
            assert r["eval"] == a["evalue"] and r["alnLength"] == a["alnLen"] and abs(r["seqId"] - a["seqId"]) == 0
            assert bt == cigs[k], (qi, k, go, ge)
        accepted += len(want)
    on_device, total = s.backtrace_counts()
    if device_backtrace == 1:
        assert total > 0 and on_device >= 0.95 * total, (on_device, total)
    elif device_backtrace == 0:
        assert on_device == 0
    s.close()
    return accepted


@pytest.mark.parametrize("dev", [0, 1])
@pytest.mark.parametrize("atype", [0, 2])
def test_structure_alignment_of_real_hit_lists_equals_reference_at_100k(world, atype, dev):
    """prefilter hit lists (1000 targets each, 50 planted homologs among them) through the batch path (k_sw2 forward over all
    pairs, reversed over the gate survivors, host gates, block-aligner backtrace): forward score / end positions of EVERY pair
    and the complete accepted records (scores, e-value bits, start/end, CIGAR, order) against the reference's alignStructure"""
    if world["ref"] is None:
        pytest.skip("oracle/_ref not built")
    accepted = _align_against_reference(world, atype, 10, 1, list(range(len(world["q3"]))), 1000, dev)
    assert accepted >= 150            # the planted homologs are found and aligned, not just random pairs


@pytest.mark.parametrize("dev", [0, 1])
@pytest.mark.parametrize("go,ge", [(8, 2), (15, 3), (3, 1), (2, 1), (25, 1)])
def test_structure_alignment_with_other_gap_costs_equals_reference(world, go, ge, dev):
    """--gap-open / --gap-extend other than 10 / 1 (any gapOpen > gapExtend >= 1 is on the device path): the same comparison against the
    reference's alignStructure run with those costs -- SW kernels (single tile and the 777-residue row-tiled query), gates, backtraces"""
    if world["ref"] is None:
        pytest.skip("oracle/_ref not built")
    accepted = _align_against_reference(world, 2, go, ge, [0, 1, 3], 300, dev)
    assert accepted >= 60


@pytest.mark.parametrize("go,ge", [(11, 0), (5, 5), (3, 7)])
def test_gap_costs_the_path_does_not_cover_are_refused(world, go, ge):
    """gap extend 0 kills the reference in the block aligner's assertion, gapOpen <= gapExtend is outside what the device SW reproduces:
    an error code and a message, not an abort"""
    par = api.default_params()
    par.gapOpen, par.gapExtend = go, ge
    s = api.Search(world["ctx"], par)
    with pytest.raises(RuntimeError, match="gap costs must satisfy"):
        s.align(world["qa"][0], world["q3"][0], np.arange(10, dtype=np.uint32))
    with pytest.raises(RuntimeError, match="gap costs must satisfy"):
        s.align_batch([world["qa"][0]], [world["q3"][0]], [np.arange(10, dtype=np.uint32)])
    s.close()


def test_multi_query_scan_equals_single_query_scans(world):
    """fsgpu_gapless_scan_multi (queries of one register class share ONE k_gapless launch: per-query work queues, score slices,
    batched selection passes) == one fsgpu_gapless_scan per query: complete score vectors and hit lists, 100k targets"""
    ctx = world["ctx"]
    more3, _ = synth.make_queries(11, seed=5, lo=250, hi=300)            # R = 16..19: several queries per launch
    qs = list(world["q3"]) + more3 + [more3[0][:33], more3[1][:16], more3[2][:1]]
    # short queries of one 16-row class run two to a kernel (PAIRED): three of class 7 (one pair + one alone), two of class 3, two of
    # class 16 with different lengths, four of class 1
    qs += [more3[3][:100], more3[4][:101], more3[5][:112], more3[6][:40], more3[7][:45], more3[8][:241], more3[9][:256],
           more3[0][:2], more3[1][:7], more3[2][:16], more3[3][:15]]
    ident = np.full(len(qs), -1, np.int64)
    ident[2], ident[6] = 777, 31415
    s = api.Search(ctx)
    single, scores = [], []
    for i, q in enumerate(qs):
        single.append(s.prefilter(q, identity=int(ident[i])))
        scores.append(ctx.gapless_scores())
    multi = s.prefilter_batch(qs, identity=ident)
    launches, batched = ctx.gapless_last_batch()
    short = [i for i, q in enumerate(qs) if len(q) <= 896]           # one-piece queries share launches per register class
    n_tiled = len(qs) - len(short)                                   # a row-tiled query runs on its own and counts as one more scan of the call
    assert batched == len(qs) and n_tiled == sum(1 for q in world["q3"] if len(q) > 896) >= 1
    launches -= n_tiled
    members = {}
    for i in short:
        members[(len(qs[i]) + 15) // 16] = members.get((len(qs[i]) + 15) // 16, 0) + 1
    assert launches == sum((1 + m % 2) if (c <= 16 and m >= 2) else 1 for c, m in members.items()) < len(short)
    assert sum(1 for c, m in members.items() if c <= 16 and m >= 2) >= 4
    for i in range(len(qs)):
        assert len(multi[i]) == len(single[i]) and (multi[i] == single[i]).all(), i
        if i in short:
            assert (ctx.gapless_scores_multi(i) == scores[i]).all(), i
    # a second batch in another order: slices / queues are reused
    order = [5, 0, 9, 2]
    again = s.prefilter_batch([qs[i] for i in order], identity=ident[order])
    for k, i in enumerate(order):
        assert (again[k] == single[i]).all()
    s.close()


def test_backtrace_pool_does_not_change_results(world):
    """the accepted hits' backtraces run on a host worker pool (search.cpp HostPool); any pool size gives the same records"""
    db, q3, qa = world["db"], world["q3"], world["qa"]
    par = api.default_params()
    par.addBacktrace = 1
    s = api.Search(world["ctx"], par)
    hit_lists = [s.prefilter(q)["id"] for q in q3]
    before = api.host_workers()
    outs = []
    for w in (0, 1, 7):
        api.set_host_workers(w)
        assert api.host_workers() == w
        res, bts = s.align_batch(qa, q3, hit_lists, with_backtrace=True)
        outs.append(([r.tobytes() for r in res], bts))
        one, bt1 = s.align(qa[0], q3[0], hit_lists[0], with_backtrace=True)
        assert one.tobytes() == res[0].tobytes() and bt1 == bts[0]
    api.set_host_workers(before)
    assert outs[0] == outs[1] == outs[2]
    assert sum(len(b) for b in outs[0][1]) >= 150
    s.close()


@pytest.fixture(scope="module")
def world1m():
    """configs[2] / configs[3] size: 1M targets, 350 M residues, 50 planted homologs for each of three queries"""
    q3, qa = synth.make_queries(3, seed=99, lo=250, hi=450)
    db = synth.make_db_fast(1000000, (q3, qa), seed=31337, homologs_per_query=50)
    ctx = api.Context(0)
    ctx.load_db(db)
    yield dict(db=db, q3=q3, qa=qa, ctx=ctx, ref=oracle_lib.load_ref())
    ctx.close()


def test_gapless_properties_at_1M_targets(world1m):
    """configs[2] size (1M targets, 350 M residues).  The reference needs ~0.3 s per query here even on 16 cores, so: a
    3000-target sample spread over the whole length range is checked against it, the full score vector through properties
    (every score in [0, cap], planted homologs score above the random background, hit list = exact top-1000 of the vector
    in (score desc, id asc) order, a second run is identical: the atomic queue order must not leak into results)."""
    q3, db, ctx, ref = world1m["q3"], world1m["db"], world1m["ctx"], world1m["ref"]
    s = api.Search(ctx)
    for q in q3:
        hits = s.prefilter(q)
        got = ctx.gapless_scores().astype(np.int32)
        hits2 = s.prefilter(q)
        assert (ctx.gapless_scores().astype(np.int32) == got).all() and (hits2 == hits).all()
        sel = helpers.o_prefilter_select(got, 30, -1, 1000)
        assert (hits["id"] == sel["key"]).all() and (hits["score"] == sel["score"]).all()
        assert got.min() >= 0 and got.max() <= 255
        assert (np.sort(got)[-40:] > np.percentile(got, 99.9)).all()
        if ref is not None:
            idx = np.linspace(0, db.n - 1, 3000).astype(np.int64)
            want = np.zeros(len(idx), np.int32)
            ref.ref_ungapped(q, len(q), 1, 0.15, db.data3di, np.ascontiguousarray(db.offsets[:-1][idx]), np.ascontiguousarray(db.lengths[idx]),
                             len(idx), 16, want)
            assert (got[idx] == want).all()
            top = np.ascontiguousarray(hits["id"][:300].astype(np.int64))
            want = np.zeros(len(top), np.int32)
            ref.ref_ungapped(q, len(q), 1, 0.15, db.data3di, np.ascontiguousarray(db.offsets[:-1][top]), np.ascontiguousarray(db.lengths[top]),
                             len(top), 16, want)
            assert (hits["score"][:300] == want).all()
    s.close()


@pytest.mark.parametrize("dev", [0, 1])
@pytest.mark.parametrize("atype", [0, 2])
def test_structure_alignment_of_real_hit_lists_equals_reference_at_1M(world1m, atype, dev):
    """configs[2] (--alignment-type 0) and configs[3] (--alignment-type 2, 3Di + AA) at their stated size: the 3 x 1000 prefilter hit lists of
    the 1M-target database through the batch path, the COMPLETE alignStructure records (forward / reverse scores, e-value bits, start / end
    positions, alignment length, identity, CIGAR, output order) against the compiled reference's alignStructure on the same pairs"""
    if world1m["ref"] is None:
        pytest.skip("oracle/_ref not built")
    accepted = _align_against_reference(world1m, atype, 10, 1, [0, 1, 2], 1000, dev)
    assert accepted >= 120            # the planted homologs are among the accepted records
