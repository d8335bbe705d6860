"""CPU tests of the module layer that need no GPU: makepaddedseqdb against the padded layout, usage errors."""
import ctypes as C
import os
import numpy as np
from foldseek_amd import api, synth, dbio


def _call(fn, args):
    arr = (C.c_char_p * len(args))(*[a.encode() for a in args])
    f = getattr(C.CDLL(api.LIB_PATH), fn)
    f.restype = C.c_int
    return f(len(args), arr)


def test_makepaddedseqdb_layout(tmp_path):
    rng = np.random.default_rng(8)
    lens = [5, 33, 8, 8, 100, 1, 64, 33]
    seqs = [rng.integers(0, 21, size=l).astype(np.uint8) for l in lens]
    masks = [rng.random(l) < 0.2 for l in lens]
    keys = [10, 11, 12, 13, 14, 15, 16, 17]
    src = str(tmp_path / "db_ss")
    dbio.write_seq_db(src, seqs, keys, masks)
    dst = str(tmp_path / "db_ss_pad")
    assert _call("fsmod_makepaddedseqdb", [src, dst]) == 0
    t, data, offsets, lengths = dbio.read_padded_db(dst)
    assert t & 0xffff == 0 and (t >> 16) & 8
    # ascending length; ties: the reference iterates a (length desc, id asc) order backwards -> id descending
    order = sorted(range(len(lens)), key=lambda i: (lens[i], -i))
    assert list(lengths) == [lens[i] for i in order]
    assert (offsets % 4 == 0).all()
    for new, old in enumerate(order):
        got = data[offsets[new]:offsets[new] + lengths[new]]
        want = seqs[old] + 32 * masks[old].astype(np.uint8)
        assert (got == want).all()
        pad = data[offsets[new] + lengths[new]:offsets[new + 1]]
        assert (pad == 20).all()
    lookup = [l.split() for l in open(dst + ".lookup")]
    assert [int(x[1]) for x in lookup] == [keys[i] for i in order]


def test_module_usage_errors(tmp_path):
    assert _call("fsmod_ungappedprefilter", ["only-one-arg"]) != 0
    assert _call("fsmod_structurealign", ["a", "b"]) != 0
    assert _call("fsmod_makepaddedseqdb", [str(tmp_path / "missing"), str(tmp_path / "out")]) != 0


def test_prefilter_module_argument_errors(tmp_path):
    """option handling of the k-mer prefilter module that is decided before any device call"""
    assert _call("fsmod_prefilter", ["a", "b"]) != 0
    rng = np.random.default_rng(3)
    seqs = [rng.integers(0, 20, size=40).astype(np.uint8) for _ in range(4)]
    src = str(tmp_path / "db_ss")
    dbio.write_seq_db(src, seqs, [1, 2, 3, 4])
    out = str(tmp_path / "out")
    assert _call("fsmod_prefilter", [src, src, out, "-k", "7"]) != 0                    # only k = 6 on the device path
    assert _call("fsmod_prefilter", [src, src, out, "--diag-score", "0"]) != 0
    assert _call("fsmod_prefilter", [src, src, out, "--exact-kmer-matching", "1"]) != 0
    assert _call("fsmod_prefilter", [str(tmp_path / "missing"), src, out]) != 0
    assert not os.path.exists(out + ".index")


def test_kmer_threshold_and_query_prepare():
    """Prefiltering::getKmerThreshold values and the per-position thresholds / ungapped profile against the oracle's
    arithmetic (bias rounding rules of QueryMatcher.cpp:267-268 and UngappedAlignment.cpp:399-403)."""
    import helpers
    assert api.kmer_threshold(9.5, 6) == 78 and api.kmer_threshold(7.5, 6) == 96 and api.kmer_threshold(9.5, 7) == 90
    m8, m2 = api.Matrix(0, 8.0, -0.2), api.Matrix(0, 2.0, -0.2)
    ksub, pb = helpers.o_submat("MAT3DI", 8.0, -0.2)
    usub, _ = helpers.o_submat("MAT3DI", 2.0, -0.2)
    rng = np.random.default_rng(12)
    q = rng.integers(0, 21, size=133).astype(np.uint8)
    seq, thr, prof = api.kmer_query_prepare(m8, m2, q, comp_bias=True, scale=0.15, kmer_thr=78)
    cbf, _ = helpers.o_round_bias(ksub, pb, q, 0.15)
    pos = [0, 1, 3, 5, 8, 9]
    assert len(thr) == len(q) - 9
    for i in range(len(thr)):
        bc = np.float32(0)
        for z in pos:
            bc = np.float32(bc + cbf[i + z])
        b = int(np.trunc(np.float64(bc) - 0.5)) if bc < 0 else int(np.trunc(np.float64(bc) + 0.5))
        assert thr[i] == max(78 - b, 0)
    us = usub.reshape(21, 21)
    for p in range(len(q)):
        c = np.float32(cbf[p])
        c4 = np.float32(np.float64(np.float32(c / np.float32(4))) + (-0.5 if c < 0 else 0.5))
        assert (prof[p] == (us[q[p]] + int(np.trunc(c4))).astype(np.int8)).all()
    _, thr0, prof0 = api.kmer_query_prepare(m8, m2, q, comp_bias=False)
    assert (thr0 == 78).all() and (prof0 == us[q].astype(np.int8)).all()
    assert len(api.kmer_query_prepare(m8, m2, q[:7])[1]) == 0
