"""Device block-aligner backtrace (k_btrace.hpp, fsgpu_block_backtrace; SURVEY row a16 on the MI355X) against the host restatement (host/block_aligner.cpp,
which the CPU suite holds to the crate's known answers and to the independent trajectory model): the SAME align_batch call with the device path on and
off must return identical records -- start positions, alignment length, sequence identity, backtrace strings -- for every accepted hit, and the device
must have answered most of them itself (hits whose block wants to grow beyond 128 rows are handed back to the host path by design)."""
import os

import numpy as np
import pytest

from foldseek_amd import api, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def world():
    q3, qa = synth.make_queries(20, seed=505, lo=40, hi=900)
    # a few long ones and two short ones: blocks of 32, 64 and 128 rows, row / column shifts, grow and shrink
    q3[0], qa[0] = q3[0][:35], qa[0][:35]
    q3[1], qa[1] = np.concatenate([q3[1], q3[2]])[:1500], np.concatenate([qa[1], qa[2]])[:1500]
    db = synth.make_db_fast(30000, (q3, qa), seed=606, homologs_per_query=40)
    ctx = api.Context(0)
    ctx.load_db(db)
    yield dict(db=db, q3=q3, qa=qa, ctx=ctx)
    ctx.close()


def _run(world, atype, go, ge, device):
    par = api.default_params()
    par.alignmentType = atype
    par.addBacktrace = 1
    par.gapOpen, par.gapExtend = go, ge
    os.environ.pop("FSGPU_DEVICE_BACKTRACE", None)
    os.environ.pop("FSGPU_BT_SHARE_MIN", None)
    if device is None:
        os.environ["FSGPU_DEVICE_BACKTRACE"] = "2"        # host pool and device share the batch's list (the default on hosts with <= 4 cores; from 1024 hits on, here from 128)
        os.environ["FSGPU_BT_SHARE_MIN"] = "128"
    else:
        os.environ["FSGPU_DEVICE_BACKTRACE"] = "1" if device else "0"
    try:
        pre = api.Search(world["ctx"])
        hits = [pre.prefilter(q)["id"][:300] for q in world["q3"]]
        pre.close()
        s = api.Search(world["ctx"], par)
        res, bts = s.align_batch(world["qa"], world["q3"], hits, with_backtrace=True)
        counts = s.backtrace_counts()
        s.close()
    finally:
        os.environ.pop("FSGPU_DEVICE_BACKTRACE", None)
        os.environ.pop("FSGPU_BT_SHARE_MIN", None)
    return res, bts, counts


@pytest.mark.parametrize("atype", [0, 2])
def test_shared_backtrace_equals_host_backtrace(world, atype):
    """FSGPU_DEVICE_BACKTRACE=2 (the default where cores are few): the pool's threads take hits from the front of the batch's list, the device aligner chunks from its back;
    both must have answered hits, and the records must be the host aligner's"""
    rs, bs, (on_dev, total) = _run(world, atype, 10, 1, None)
    rh, bh, (on_dev_h, total_h) = _run(world, atype, 10, 1, False)
    assert on_dev_h == 0 and total_h == total and total >= 400
    assert 0 < on_dev < total, (on_dev, total)
    for q in range(len(rs)):
        assert rs[q].tobytes() == rh[q].tobytes(), q
        assert bs[q] == bh[q], q


@pytest.mark.parametrize("atype,go,ge", [(0, 10, 1), (2, 10, 1), (2, 8, 2), (2, 3, 1), (0, 15, 3)])
def test_device_backtrace_equals_host_backtrace(world, atype, go, ge):
    rd, bd, (on_dev, total) = _run(world, atype, go, ge, True)
    rh, bh, (on_dev_h, total_h) = _run(world, atype, go, ge, False)
    assert on_dev_h == 0 and total_h == total and total >= 400
    assert on_dev >= 0.8 * total, (on_dev, total)                      # the device answers most hits itself
    n = 0
    for q in range(len(rd)):
        assert len(rd[q]) == len(rh[q]), q
        assert rd[q].tobytes() == rh[q].tobytes(), q
        assert bd[q] == bh[q], q
        n += len(rd[q])
    assert n >= 400


def test_default_choice_follows_the_cores_the_process_may_use(world):
    """FSGPU_DEVICE_BACKTRACE unset: a process that may run on at most 4 CPUs (a rank pinned to its share of the node's cores) gets its backtraces from the
    device aligner, one with more from the host pool (precomputeBacktraces; measured in profiles/r06_backtrace_modes.txt)"""
    if not hasattr(os, "sched_setaffinity"):
        pytest.skip("no sched_setaffinity on this platform")
    cpus = sorted(os.sched_getaffinity(0))
    os.environ.pop("FSGPU_DEVICE_BACKTRACE", None)
    par = api.default_params()
    par.alignmentType = 2
    par.addBacktrace = 1
    pre = api.Search(world["ctx"])
    hits = [pre.prefilter(q)["id"][:300] for q in world["q3"]]
    pre.close()
    out = {}
    try:
        for name, mask in (("two", set(cpus[:2])), ("all", set(cpus))):
            os.sched_setaffinity(0, mask)
            s = api.Search(world["ctx"], par)
            res, bts = s.align_batch(world["qa"], world["q3"], hits, with_backtrace=True)
            out[name] = (res, bts, s.backtrace_counts())
            s.close()
    finally:
        os.sched_setaffinity(0, set(cpus))
    (r2, b2, (dev2, tot2)), (ra, ba, (deva, tota)) = out["two"], out["all"]
    assert tot2 == tota and tot2 >= 400
    assert dev2 >= 0.8 * tot2, (dev2, tot2)                     # two CPUs: the device aligner
    if len(cpus) > 4 and api.lib().fshost_usable_cores() > 4:
        assert deva == 0, (deva, tota)                          # the box's cores: the host pool
    for q in range(len(r2)):
        assert r2[q].tobytes() == ra[q].tobytes() and b2[q] == ba[q], q


def test_long_gapped_pairs_fall_back_and_still_agree(world):
    """targets with a long insertion: the aligner's block grows beyond what the device keeps in LDS; those hits come back through the host path, the
    records stay identical"""
    rng = np.random.default_rng(9)
    q3, qa = world["q3"][3], world["qa"][3]
    db0 = world["db"]
    # a small second database: the query with 150-residue insertions at several places
    t3, ta = [], []
    for k in range(40):
        a = int(rng.integers(10, len(q3) - 10))
        ins3 = rng.integers(0, 20, 150 + 5 * k).astype(np.uint8); insa = rng.integers(0, 20, 150 + 5 * k).astype(np.uint8)
        t3.append(np.concatenate([q3[:a], ins3, q3[a:]])); ta.append(np.concatenate([qa[:a], insa, qa[a:]]))
    db = synth.make_db_fast(2000, ([q3], [qa]), seed=707, homologs_per_query=10)
    # overwrite the first 40 long-enough entries with the constructed targets
    done = 0
    for i in range(db.n - 1, -1, -1):
        if done == len(t3):
            break
        L = int(db.lengths[i])
        if L >= len(t3[done]):
            o = int(db.offsets[i])
            db.data3di[o:o + len(t3[done])] = t3[done]; db.dataaa[o:o + len(ta[done])] = ta[done]
            done += 1
    ctx = api.Context(0)
    ctx.load_db(db)
    par = api.default_params()
    par.alignmentType = 2
    par.addBacktrace = 1
    out = []
    for dev in (True, False):
        os.environ["FSGPU_DEVICE_BACKTRACE"] = "1" if dev else "0"
        os.environ["FSGPU_BT_PASS2"] = "1"          # the second pass (blocks of up to 512 rows) takes what the first hands back, however few
        try:
            s = api.Search(ctx, par)
            res, bts = s.align_batch([qa], [q3], [np.arange(db.n, dtype=np.uint32)], with_backtrace=True)
            out.append((res[0], bts[0], s.backtrace_counts()))
            s.close()
        finally:
            os.environ.pop("FSGPU_DEVICE_BACKTRACE", None)
            os.environ.pop("FSGPU_BT_PASS2", None)
    ctx.close()
    (rd, bd, cd), (rh, bh, ch) = out
    assert rd.tobytes() == rh.tobytes() and bd == bh
    assert cd[1] == ch[1] and cd[1] >= 10 and ch[0] == 0
    assert any("I" * 100 in b or "D" * 100 in b for b in bd)           # the long gaps are in the answers
