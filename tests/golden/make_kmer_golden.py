#!/usr/bin/env python3
"""Generates tests/golden/kmer_v1.npz from the REFERENCE's own k-mer prefilter classes (oracle/_ref/libfsref.so built by
oracle/Makefile from the sources under /root/reference; driver oracle/ref_kmer_harness.cpp).  Build container only:
    make -C oracle && python tests/golden/make_kmer_golden.py
Frozen for a seeded synthetic 3Di database (padded layout, soft-mask flags kept): the prefilter's two substitution
matrices, sample rows of the extended 3-mer matrix, sample similar-k-mer lists, sample index lists + the offset table's
checksum, and the complete QueryMatcher::matchQuery hit lists (id, score, diagonal, order) of several queries under the
parameter sets that change the arrival-order rules (max-seqs truncation, BINSIZE, databaseHits refills, identity)."""
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from foldseek_amd import synth          # noqa: E402
import kmer_lib as K                    # noqa: E402

R = K.load_ref()
assert R is not None, "build oracle/_ref first"
NQ, N = 5, 1200
q3, qa = synth.make_queries(NQ, seed=42, hi=700)
q3.append(np.array([20] * 12 + list(np.random.default_rng(3).integers(0, 20, 60)), np.uint8))    # leading X run
db = synth.make_db(N, (q3, qa + [qa[0][:72]]), seed=7, homologs_per_query=20, mask_frac=0.03, hi=900)
targets = [db.seq(i, "3di", unmask=False) for i in range(db.n)]
r = K.RefKpf(R, targets, threads=8)
out = dict(db_data3di=db.data3di, db_offsets=db.offsets, db_lengths=db.lengths,
           q_lens=np.array([len(q) for q in q3]), q3=np.concatenate(q3),
           sub_kmer=r.submat(0), sub_ungapped=r.submat(1), l2=np.array([2 * 1024 * 1024], np.uint64))
rows = [0, 19, 400, 4321, 7999]
out["rows"] = np.array(rows)
out["row_scores"] = np.stack([r.row(3, i)[0] for i in rows])
out["row_index"] = np.stack([r.row(3, i)[1] for i in rows])
rng = np.random.default_rng(11)
kms = rng.integers(0, 20, (12, 6)).astype(np.uint8)
thr = rng.integers(50, 110, 12)
lists = [r.kmer_list(kms[i], int(thr[i])) for i in range(12)]
out["kl_kmers"], out["kl_thr"] = kms, thr
out["kl_len"] = np.array([len(x) for x in lists])
out["kl_cat"] = np.concatenate(lists)
off = r.offsets()
out["index_entries"] = np.array([off[-1]], np.uint64)
out["index_offsets_sum"] = np.array([np.bitwise_xor.reduce(off * np.arange(1, len(off) + 1, dtype=np.uint64))], np.uint64)
nz = np.nonzero(np.diff(off.astype(np.int64)))[0]
pick = rng.choice(nz, 64, replace=False)
il = [r.index_list(k) for k in pick]
out["il_kmers"] = pick
out["il_len"] = np.array([len(x[0]) for x in il])
out["il_seq"] = np.concatenate([x[0] for x in il])
out["il_pos"] = np.concatenate([x[1] for x in il])
VARIANTS = [dict(), dict(maxResListLen=40), dict(maxResListLen=40, bins=8), dict(maxResListLen=3),
            dict(maxResListLen=200, maxDbMatches=6000), dict(maxResListLen=60, maxDbMatches=2500, bins=4),
            dict(compBias=0, maxResListLen=25, maxDbMatches=5000), dict(minDiagScoreThr=12, maxResListLen=1500)]
ident = np.array([-1, 5, -1, 77, -1, -1], np.int64)
out["identity"] = ident
out["n_variants"] = np.array([len(VARIANTS)])
for vi, kw in enumerate(VARIANTS):
    base = dict(maxResListLen=1000, bins=2, maxDbMatches=0, foundDiagonalsSize=0, compBias=1, minDiagScoreThr=30)
    base.update(kw)
    r.set(**base)
    rr, rs, _ = r.run(q3, ident)
    out[f"v{vi}_params"] = np.array([base["maxResListLen"], base["bins"], base["maxDbMatches"], base["foundDiagonalsSize"], base["compBias"], base["minDiagScoreThr"]], np.int64)
    out[f"v{vi}_cnt"] = np.array([len(x) for x in rr])
    out[f"v{vi}_hits"] = np.concatenate(rr) if sum(len(x) for x in rr) else np.zeros(0, K.HIT_DT)
    out[f"v{vi}_stats"] = rs
    print(vi, kw, [len(x) for x in rr], "overflow", rs[:, 2].tolist())
np.savez_compressed(os.path.join(HERE, "kmer_v1.npz"), **out)
print("wrote", os.path.join(HERE, "kmer_v1.npz"), os.path.getsize(os.path.join(HERE, "kmer_v1.npz")), "bytes")
