"""CPU tests (no GPU): host-side C++ of libfsgpu.so against the oracle, C-ABI symbol export, formats."""
import ctypes as C
import os
import re
import numpy as np
import pytest

from foldseek_amd import api, synth
import helpers
from oracle_lib import load_oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    L = C.CDLL(api.LIB_PATH)
    for hdr in ("fsgpu.h", "fshost.h"):
        txt = open(os.path.join(ROOT, "include", hdr)).read()
        names = set(re.findall(r"\b(fs(?:gpu|host)_[a-z0-9_]+)\s*\(", txt))
        assert names, hdr
        for n in names:
            assert hasattr(L, n), f"{n} declared in include/{hdr} but not exported by libfsgpu.so"


@pytest.mark.parametrize("which,name,bf", [(0, "MAT3DI", 2.0), (0, "MAT3DI", 2.1), (0, "MAT3DI", 8.0), (1, "BLOSUM62", 1.4),
                                           (1, "BLOSUM62", 0.0)])
def test_matrix_matches_oracle(which, name, bf):
    m = api.Matrix(which, bf, 0.0)
    sub, pb = helpers.o_submat(name, bf)
    assert (m.scores().ravel() == sub).all()
    assert (m.background() == pb).all()


def test_matrix_from_text_roundtrip():
    # rebuild the .out text from the parameter table and parse it back through the user-matrix path
    import fsparams
    p = fsparams.load()["MAT3DI"]
    letters = "ACDEFGHIKLMNPQRSTVWYX"
    txt = "# 3Di\n# Background (precomputed optional): " + " ".join(repr(float(x)) for x in p["back"]) + "\n"
    txt += "# Lambda     (precomputed optional): " + repr(float(p["lam"])) + "\n"
    txt += "    " + "   ".join(letters) + "\n"
    for i, a in enumerate(letters):
        txt += a + " " + " ".join(repr(float(x)) for x in p["score"][i]) + "\n"
    m = api.Matrix(text=txt, bit_factor=2.0)
    assert (m.scores() == api.Matrix(0, 2.0).scores()).all()


def test_letter_mapping():
    m = api.Matrix(1, 1.4)
    codes = m.encode("ACDEFGHIKLMNPQRSTVWYXacdjzbuo*")
    assert list(codes[:21]) == list(range(21))
    assert list(codes[21:24]) == [0, 1, 2]
    assert codes[24] == 9 and codes[25] == 3 and codes[26] == 2      # J->L, Z->E, B->D
    assert (codes[27:] == 20).all()                                  # U, O, * -> X


@pytest.mark.parametrize("L", [1, 7, 39, 40, 41, 350, 1500])
def test_comp_bias_matches_oracle(L):
    rng = np.random.default_rng(L)
    for which, name, bf, scale in ((0, "MAT3DI", 2.0, 0.15), (1, "BLOSUM62", 1.4, 1.0), (1, "BLOSUM62", 1.4, 0.5)):
        seq = rng.integers(0, 21, size=L).astype(np.uint8)
        m = api.Matrix(which, bf)
        sub, pb = helpers.o_submat(name, bf)
        cbf, _ = helpers.o_round_bias(sub, pb, seq, scale)
        assert (m.comp_bias(seq, scale) == cbf).all()


def test_prefilter_profile_matches_oracle():
    O = load_oracle()
    rng = np.random.default_rng(3)
    m = api.Matrix(0, 2.0)
    sub, pb = helpers.o_submat("MAT3DI", 2.0)
    for L in (5, 64, 350, 777):
        q = rng.integers(0, 21, size=L).astype(np.uint8)
        for cbon in (True, False):
            pssm, cap = api.prefilter_profile(m, q, cbon, 0.15)
            cb = helpers.o_round_bias(sub, pb, q, 0.15)[1] if cbon else np.zeros(L, np.int8)
            bias = O.fso_ungapped_bias(sub.astype(np.int8), 21, cb, L)
            assert cap == 255 - bias
            ref = sub.reshape(21, 21)[:, q].astype(np.int32) + cb.astype(np.int32)[None, :]
            assert (pssm.astype(np.int32) == ref).all()


@pytest.mark.parametrize("atype", [0, 2])
def test_align_profiles_match_oracle(atype):
    rng = np.random.default_rng(11 + atype)
    mAA = api.Matrix(1, 1.4 if atype == 2 else 0.0)
    m3 = api.Matrix(0, 2.1)
    for L in (3, 100, 350, 1200):
        qa = rng.integers(0, 21, size=L).astype(np.uint8)
        q3 = rng.integers(0, 21, size=L).astype(np.uint8)
        pA, p3, cbA, cbS = api.align_profiles(mAA, m3, qa, q3, True, 0.5)
        oA, o3, ocA, ocS = helpers.o_align_profiles(qa, q3, atype)
        assert (pA.ravel() == oA).all() and (p3.ravel() == o3).all()
        assert (cbA == ocA).all() and (cbS == ocS).all()
        if atype == 0:
            assert not pA.any()


def test_evalue_network_matches_oracle():
    O = load_oracle()
    nn = np.fromfile(os.path.join(ROOT, "foldseek_amd", "data", "evalue_nn.bin"), dtype=np.uint8)
    rng = np.random.default_rng(5)
    ev = api.Evaluer(35_000_000)
    for L in (30, 350, 2000):
        q = rng.integers(0, 21, size=L).astype(np.uint8)
        lam, mu = ev.mu_lambda(q)
        a, b = C.c_double(), C.c_double()
        O.fso_predict_mu_lambda(nn, q, L, 21, C.byref(a), C.byref(b))
        assert (lam, mu) == (a.value, b.value)
        for score in (-20, 0, 31, 80, 250, 3000):
            assert ev.evalue_corr(score, lam, mu) == O.fso_evalue_corr(score, lam, mu, np.log(35_000_000.0))


def test_formats():
    assert api.format_prefilter_hit(4711, 238, 0) == "4711\t238\t0\n"
    assert api.format_prefilter_hit(0, 31, 65535) == "0\t31\t-1\n"
    r = np.zeros(1, api.RESULT_DT)
    r["dbKey"], r["score"], r["seqId"], r["eval"] = 12, 345, 0.0561, 1.234e-7
    r["qStartPos"], r["qEndPos"], r["qLen"], r["dbStartPos"], r["dbEndPos"], r["dbLen"] = 0, 99, 100, 3, 104, 120
    buf = C.create_string_buffer(4096)
    n = api.lib().fshost_format_result(buf, C.c_void_p(r.ctypes.data), b"MMMIIMDDM", 1)
    assert buf.raw[:n].decode() == "12\t345\t0.056\t1.234E-07\t0\t99\t100\t3\t104\t120\t3M2I1M2D1M\n"
    r["seqId"] = 1.0
    n = api.lib().fshost_format_result(buf, C.c_void_p(r.ctypes.data), None, 0)
    assert buf.raw[:n].decode().split("\t")[2] == "1.00"      # the reference's own off-by-one (Util.cpp:252-263 + Matcher.cpp:289), pinned by tests/golden/scop_v1
    r["seqId"] = 0.5
    n = api.lib().fshost_format_result(buf, C.c_void_p(r.ctypes.data), None, 0)
    assert buf.raw[:n].decode().split("\t")[2] == "0.500"


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(api.FsgpuError):
        api.Context(0)


def test_gapless_work_list_planner():
    """fsgpu_gapless_plan_items (host only): column segments of long stripes.  Invariants: the segments of a stripe tile its
    columns exactly once with their own ("fresh") parts, every later segment starts `overlap` chunks before its fresh part,
    unsplit stripes appear whole, items are ordered longest first, no split without need, and the chosen cut is never worse
    than not cutting under the planner's own cost model."""
    import ctypes as C
    L = api.lib()
    rng = np.random.default_rng(12)

    def plan(lens, ov, waves):
        lens = np.ascontiguousarray(lens, np.uint32)
        cap = C.c_uint32(0)
        n = L.fsgpu_gapless_plan_items(lens.ctypes.data, len(lens), ov, float(waves), None, 0, C.byref(cap))
        items = np.zeros(max(1, n), np.uint64)
        assert L.fsgpu_gapless_plan_items(lens.ctypes.data, len(lens), ov, float(waves), items.ctypes.data, n, None) == n
        return items[:n], int(cap.value)

    for trial in range(40):
        n = int(rng.integers(1, 3000))
        lens = np.clip(np.rint(rng.gamma(2.2, 22.0 / 2.2, size=n)), 0 if trial % 5 == 0 else 1, 125).astype(np.uint32)
        ov = int(rng.choice([0, 1, 7, 16, 22, 32]))
        waves = float(rng.choice([8, 64, 3072, 4096]))
        items, cap = plan(lens, ov, waves)
        stripe = (items >> np.uint64(32)).astype(np.int64)
        split = ((items >> np.uint64(31)) & np.uint64(1)).astype(bool)
        b0 = ((items >> np.uint64(16)) & np.uint64(0x7fff)).astype(np.int64)
        e = (items & np.uint64(0xffff)).astype(np.int64)
        size = e - b0
        assert (size > 0).all() and (np.diff(size) <= 0).all()                      # longest first
        assert set(stripe.tolist()) == set(np.flatnonzero(lens > 0).tolist())      # empty stripes are dropped, nothing else
        total_fresh = 0
        for s_ in np.unique(stripe):
            m = stripe == s_
            Ls = int(lens[s_])
            if not split[m].any():
                assert m.sum() == 1 and b0[m][0] == 0 and e[m][0] == Ls and (Ls <= cap or ov == 0)
                total_fresh += Ls
                continue
            assert split[m].all() and ov > 0 and Ls > cap
            order = np.argsort(e[m])
            ee, bb = e[m][order], b0[m][order]
            fresh_begin = np.concatenate(([0], ee[:-1]))                          # a segment's own columns start where the previous one ended
            assert ee[-1] == Ls and (np.diff(ee) > 0).all()
            assert (bb == np.maximum(0, fresh_begin - ov)).all()                    # warm-up = overlap chunks (clipped at the stripe start)
            assert (ee - bb <= cap + 1).all()                                        # no item longer than the cut (+ rounding of the equal split)
            total_fresh += int((ee - fresh_begin).sum())
        assert total_fresh == int(lens.sum())
        if ov == 0 or lens.max(initial=0) <= 2 * ov:
            assert not split.any() and cap == lens.max(initial=0)
        # cost model: max(longest item, work / waves) must not be worse than the uncut list
        cost = max(float(size.max(initial=0)), float(size.sum()) / waves)
        uncut = max(float(lens.max(initial=0)), float(lens.sum()) / waves)
        assert cost <= uncut + 1.0
    # the bench DB shape: only the handful of longest stripes is cut, the extra work stays below 1 %
    lens = np.sort(np.clip(np.rint(rng.gamma(2.2, 350.0 / 2.2, size=100000)), 30, 2000))[::-1]
    s16 = ((lens.reshape(-1, 8).max(axis=1) + 15) // 16).astype(np.uint32)
    items, cap = plan(s16, 21, 3072)
    size = (items & np.uint64(0xffff)).astype(np.int64) - ((items >> np.uint64(16)) & np.uint64(0x7fff)).astype(np.int64)
    assert 80 <= cap <= 100 and size.max() <= cap + 1 and size.sum() <= 1.01 * s16.sum() and len(items) < len(s16) + 64


def test_bench_usable_cores():
    """bench.py sizes its CPU baselines and its host wait policy by the cores this process may really use"""
    import importlib, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    bench = importlib.import_module("bench")
    n = bench.usable_cores()
    assert 1 <= n <= (os.cpu_count() or 1)
    if hasattr(os, "sched_getaffinity"):
        assert n <= len(os.sched_getaffinity(0))


def test_kmer_coarse_key_planner():
    """fsgpu_kmer_plan_coarse (host only): the coarse keys of the k-mer hit-stream partition.  Invariants: key(t) = blkKey[t >> 10] is monotone in t and
    maps every target into [0, keys), keyFirst is its inverse, a key is a run of whole blocks of 1024 ids, at most 64 of them (the low 16 bits of an id
    are unique inside a key), at most 512 keys; with blocksPerKey > 0 every key but the last holds exactly that many blocks; the balanced plan (0) aims at
    128 keys of equal residue count: a key is only closed early because the next block would take it beyond a 128th of the residues, or at twice the
    average number of blocks"""
    L = api.lib()
    rng = np.random.default_rng(21)
    for trial in range(30):
        n = int(rng.integers(1, 400000)) if trial % 5 else int(rng.integers(1, 3000))
        kind = trial % 3
        if kind == 0:
            lens = np.clip(np.rint(rng.gamma(2.0, 175.0, size=n)), 1, 3000).astype(np.int32)
        elif kind == 1:
            lens = np.sort(np.clip(np.rint(rng.gamma(2.0, 175.0, size=n)), 1, 3000).astype(np.int32))[::-1].copy()      # length-sorted DB
        else:
            lens = np.full(n, int(rng.integers(1, 2000)), np.int32)
        nblk = max(1, (n + 1023) // 1024)
        for bpk in (0, 1, 2, 16, 64):
            blk = np.zeros(nblk, np.uint16)
            cap = nblk + 2
            first = np.zeros(cap, np.uint32)
            keys = L.fsgpu_kmer_plan_coarse(lens.ctypes.data, n, bpk, blk.ctypes.data, first.ctypes.data, cap)
            assert 1 <= keys <= min(nblk, 512), (n, bpk, keys)
            assert L.fsgpu_kmer_plan_coarse(lens.ctypes.data, n, bpk, None, None, cap) == keys           # sizes only
            assert L.fsgpu_kmer_plan_coarse(lens.ctypes.data, n, bpk, blk.ctypes.data, first.ctypes.data, keys) < 0   # cap too small: keys + 1 needed
            t = np.arange(n, dtype=np.int64)
            k = blk[t >> 10].astype(np.int64)
            assert k[0] == 0 and (np.diff(k) >= 0).all() and (np.diff(k) <= 1).all() and k[-1] == keys - 1
            assert first[0] == 0 and first[keys] == n and (first[k] <= t).all() and (t < first[k + 1]).all()
            assert (first[:keys] % 1024 == 0).all()
            blocks = np.bincount(blk.astype(np.int64), minlength=keys)
            assert blocks.min() >= 1 and blocks.max() <= 64 and (np.diff(first[:keys + 1].astype(np.int64)) <= 65536).all()
            if bpk:
                assert (blocks[:-1] == bpk).all() and blocks[-1] <= bpk
            else:
                res = np.add.reduceat(lens.astype(np.int64), np.arange(0, n, 1024))
                target = max(1, -(-int(lens.sum()) // 128))
                capb = min(64, max(1, 2 * -(-nblk // 128)))
                assert blocks.max() <= capb
                kres = np.bincount(blk.astype(np.int64), weights=res, minlength=keys)
                for j in range(keys - 1):           # closed because full, or because the next block did not fit
                    nxt = int(res[int(first[j + 1]) >> 10])
                    assert blocks[j] == capb or kres[j] + nxt > target, (j, blocks[j], kres[j], nxt, target)
                assert (kres[blocks > 1] <= target).all()                                  # only a single over-long block exceeds the target
    assert L.fsgpu_kmer_plan_coarse(None, 0, 0, None, None, 4) == 1                        # an empty database still has one (empty) key
    assert L.fsgpu_kmer_plan_coarse(lens.ctypes.data, n, 65, None, None, 4) < 0
