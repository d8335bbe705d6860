"""Host-only modules against the reference BINARY, no GPU needed (runs wherever oracle/_ref_full/bin/foldseek exists: this container and the
GPU box): makepaddedseqdb on a database full of edge cases -- empty entries, one-residue entries, X, soft-masked residues, lengths around the
4-byte padding, headers with database prefixes -- must write the same seven files byte for byte."""
import os
import subprocess

import numpy as np
import pytest

from foldseek_amd import dbio

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FS = os.path.join(ROOT, "oracle", "_ref_full", "bin", "foldseek")
BIN = os.path.join(ROOT, "foldseek_amd", "bin", "fsgpu-modules")
pytestmark = pytest.mark.skipif(not os.path.exists(FS), reason="oracle/_ref_full/bin/foldseek not built")


def _sorted_index(path):
    lines = sorted(open(path).read().splitlines(), key=lambda l: int(l.split()[0]))
    open(path, "w").write("\n".join(lines) + "\n")


@pytest.mark.parametrize("write_lookup", ["1", "0"])
def test_makepaddedseqdb_edge_cases_equal_the_reference_binary(tmp_path, write_lookup):
    rng = np.random.default_rng(3)
    lens = [0, 1, 2, 3, 4, 5, 17, 300, 1, 0, 64, 65, 2000, 33, 7, 8]
    seqs = [rng.integers(0, 21, L).astype(np.uint8) for L in lens]          # code 20 = X
    masks = [rng.random(L) < 0.2 for L in lens]
    keys = [5, 9, 2, 100, 7, 8, 1, 50, 51, 52, 3, 4, 6, 77, 1000000, 12]
    w = str(tmp_path)
    dbio.write_seq_db(os.path.join(w, "t_ss"), seqs, keys, masks)
    heads = ["sp|P%05d|NAME_%d some text" % (k, k) if k % 3 else ("gi|%d|ref|XP_%d.1| hypothetical" % (k, k) if k % 2 else "plain_%d words" % k) for k in keys]
    with open(os.path.join(w, "t_ss_h"), "wb") as f, open(os.path.join(w, "t_ss_h.index"), "w") as fi:
        off = 0
        for k, h in zip(keys, heads):
            b = h.encode() + b"\n\0"
            f.write(b); fi.write(f"{k}\t{off}\t{len(b)}\n"); off += len(b)
    np.array([12], np.int32).tofile(os.path.join(w, "t_ss_h.dbtype"))
    _sorted_index(os.path.join(w, "t_ss.index")); _sorted_index(os.path.join(w, "t_ss_h.index"))
    r = subprocess.run([FS, "base:makepaddedseqdb", "t_ss", "ref_pad", "--write-lookup", write_lookup, "--threads", "1", "-v", "1"], cwd=w, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-2000:]
    m = subprocess.run([BIN, "makepaddedseqdb", "t_ss", "mine_pad", "--write-lookup", write_lookup, "--threads", "1"], cwd=w, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert m.returncode == 0, m.stdout[-2000:]
    for ext in ("", ".index", ".dbtype", ".lookup", "_h", "_h.index", "_h.dbtype"):
        a, b = os.path.join(w, "ref_pad" + ext), os.path.join(w, "mine_pad" + ext)
        assert os.path.exists(a) == os.path.exists(b), ext
        if os.path.exists(a):
            assert open(a, "rb").read() == open(b, "rb").read(), ext
    assert os.path.exists(os.path.join(w, "ref_pad.lookup")) == (write_lookup == "1")
