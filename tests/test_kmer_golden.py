"""Committed golden vectors of the k-mer prefilter (tests/golden/kmer_v1.npz, generated from the reference's own compiled
classes by tests/golden/make_kmer_golden.py).  CPU part: the C oracle against the fixture.  GPU part (-m gpu): the device
pipeline through the C ABI against the fixture."""
import os
import numpy as np
import pytest

import helpers as H
import kmer_lib as K
from foldseek_amd import api, synth

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kmer_v1.npz"))


def _queries():
    lens = G["q_lens"]
    o = np.concatenate([[0], np.cumsum(lens)])
    return [np.ascontiguousarray(G["q3"][o[i]:o[i + 1]]) for i in range(len(lens))]


def _db():
    return synth.PaddedDB(np.ascontiguousarray(G["db_data3di"]), None, G["db_offsets"], G["db_lengths"])


def _variant(vi):
    p = G[f"v{vi}_params"]
    kw = dict(maxResListLen=int(p[0]), bins=int(p[1]), maxDbMatches=int(p[2]), foundDiagonalsSize=int(p[3]), compBias=int(p[4]), minDiagScoreThr=int(p[5]))
    cnt = G[f"v{vi}_cnt"]
    o = np.concatenate([[0], np.cumsum(cnt)])
    hits = [G[f"v{vi}_hits"][o[i]:o[i + 1]] for i in range(len(cnt))]
    return kw, hits, G[f"v{vi}_stats"]


def _split(cat, lens):
    o = np.concatenate([[0], np.cumsum(lens)])
    return [cat[o[i]:o[i + 1]] for i in range(len(lens))]


@pytest.fixture(scope="module")
def ora():
    db = _db()
    targets = [db.seq(i, "3di", unmask=False) for i in range(db.n)]
    ksub, pb = H.o_submat("MAT3DI", 8.0, -0.2)
    usub, _ = H.o_submat("MAT3DI", 2.0, -0.2)
    assert (ksub == G["sub_kmer"].ravel()).all() and (usub == G["sub_ungapped"].ravel()).all()
    o = K.OraKpf(K.load_ora(), ksub, pb, usub, targets)
    yield o
    o.close()


def test_oracle_pieces_against_golden(ora):
    for k, idx in enumerate(G["rows"]):
        s, ix = ora.row(3, int(idx))
        assert (s == G["row_scores"][k]).all() and (ix == G["row_index"][k]).all()
    for km, thr, want in zip(G["kl_kmers"], G["kl_thr"], _split(G["kl_cat"], G["kl_len"])):
        assert (ora.kmer_list(km, int(thr)) == want).all()
    off = ora.offsets()
    assert off[-1] == G["index_entries"][0]
    assert np.bitwise_xor.reduce(off * np.arange(1, len(off) + 1, dtype=np.uint64)) == G["index_offsets_sum"][0]
    for k, s, p in zip(G["il_kmers"], _split(G["il_seq"], G["il_len"]), _split(G["il_pos"], G["il_len"])):
        a, b = ora.index_list(int(k))
        assert (a == s).all() and (b == p).all()


@pytest.mark.parametrize("vi", range(int(G["n_variants"][0])))
def test_oracle_hits_against_golden(ora, vi):
    kw, hits, stats = _variant(vi)
    ora.set(l2CacheSize=int(G["l2"][0]), **kw)
    res, st = ora.run(_queries(), G["identity"])
    for q in range(len(hits)):
        assert len(res[q]) == len(hits[q]) and (res[q] == hits[q]).all(), (vi, q)
        assert np.allclose(st[q], stats[q])


@pytest.fixture(scope="module")
def gpu():
    ctx = api.Context(0)
    ctx.load_db(_db())
    m8, m2 = api.Matrix(0, 8.0, -0.2), api.Matrix(0, 2.0, -0.2)
    assert (m8.scores() == G["sub_kmer"]).all() and (m2.scores() == G["sub_ungapped"]).all()
    ctx.kmer_index_build(m8, kmer_thr=78)
    return ctx, m8, m2


@pytest.mark.gpu
def test_gpu_index_against_golden(gpu):
    ctx = gpu[0]
    assert ctx.kmer_index_entries == int(G["index_entries"][0])
    off, seq, pos, _ = ctx.kmer_index_reference_order(G["db_data3di"].size)
    assert np.bitwise_xor.reduce(off * np.arange(1, len(off) + 1, dtype=np.uint64)) == G["index_offsets_sum"][0]
    for k, s, p in zip(G["il_kmers"], _split(G["il_seq"], G["il_len"]), _split(G["il_pos"], G["il_len"])):
        lo, hi = int(off[int(k)]), int(off[int(k) + 1])
        assert (seq[lo:hi] == s).all() and (pos[lo:hi] == p).all()
    for k, idx in enumerate(G["rows"]):
        s, ix = ctx.kmer_row(int(idx))
        assert (s == G["row_scores"][k]).all() and (ix == G["row_index"][k]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("vi", range(int(G["n_variants"][0])))
def test_gpu_hits_against_golden(gpu, vi):
    ctx, m8, m2 = gpu
    kw, hits, stats = _variant(vi)
    prep = [api.kmer_query_prepare(m8, m2, q, comp_bias=bool(kw["compBias"]), scale=0.15, kmer_thr=78) for q in _queries()]
    res, status, st = ctx.kmer_search(prep, identity=G["identity"], max_res=kw["maxResListLen"], min_diag=kw["minDiagScoreThr"], bins=kw["bins"],
                                      max_db_matches=kw["maxDbMatches"], found_diagonals_size=kw["foundDiagonalsSize"],
                                      l2_cache_size=int(G["l2"][0]), want_stats=True)
    for q in range(len(hits)):
        assert status[q] == 0
        assert len(res[q]) == len(hits[q]) and (res[q] == hits[q]).all(), (vi, q)
        assert np.allclose(st[q], stats[q])
