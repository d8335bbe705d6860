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
