#!/usr/bin/env python3
"""Generates tests/golden/hotpath_v1.npz from the REFERENCE's own code (oracle/_ref/libfsref.so, i.e. the sources under
/root/reference compiled in this container by oracle/Makefile).  Run in the build container only:
    make -C oracle && python tests/golden/make_golden.py
The fixture freezes, for seeded synthetic inputs: the substitution matrices, composition biases, the gapless score of
every (query, target) pair, forward / reversed-query alignScoreEndPos results for both alignment types, the e-value
network outputs, and the complete alignStructure outcome (gates, start positions, backtrace, identities)."""
import ctypes as C
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from foldseek_amd import synth          # noqa: E402
from oracle_lib import load_ref, REFSW_DT, REFALN_DT   # noqa: E402

R = load_ref()
assert R is not None, "build oracle/_ref first"
rng = np.random.default_rng(2026)
lens = (33, 97, 180, 350, 520, 777)
q3 = [rng.choice(20, size=L).astype(np.uint8) for L in lens]
qa = [rng.choice(20, size=L).astype(np.uint8) for L in lens]
db = synth.make_db(240, (q3, qa), seed=99, homologs_per_query=20, mask_frac=0.03, hi=900)
t3 = np.where(db.data3di >= 32, db.data3di - 32, db.data3di).astype(np.uint8)
out = dict(db_data3di=db.data3di, db_dataaa=db.dataaa, db_offsets=db.offsets, db_lengths=db.lengths,
           q_lens=np.array(lens), q3=np.concatenate(q3), qa=np.concatenate(qa))
for which, name in ((0, "mat3di"), (1, "blosum62")):
    for bf in (2.0, 2.1, 1.4, 0.0):
        sub = np.zeros(21 * 21, np.int16)
        pb = np.zeros(21)
        R.ref_submat(which, bf, 0.0, sub, pb)
        out[f"sub_{name}_{bf}"] = sub
        out[f"pback_{name}_{bf}"] = pb
cb = []
for qi in range(len(lens)):
    a = np.zeros(lens[qi], np.float32)
    R.ref_comp_bias(0, 2.0, 0.0, q3[qi], lens[qi], 0.15, a)
    cb.append(a)
out["cb_pref"] = np.concatenate(cb)
ung = np.zeros((len(lens), db.n), np.int32)
ung_nocb = np.zeros((len(lens), db.n), np.int32)
for qi in range(len(lens)):
    R.ref_ungapped(q3[qi], lens[qi], 1, 0.15, db.data3di, db.offsets[:-1].copy(), db.lengths, db.n, 1, ung[qi])
    R.ref_ungapped(q3[qi], lens[qi], 0, 0.15, db.data3di, db.offsets[:-1].copy(), db.lengths, db.n, 1, ung_nocb[qi])
out["ungapped"] = ung
out["ungapped_nocb"] = ung_nocb
mulam = np.zeros((len(lens), 2))
for qi in range(len(lens)):
    a, b = C.c_double(), C.c_double()
    R.ref_mu_lambda(q3[qi], lens[qi], db.residues, C.byref(a), C.byref(b))
    mulam[qi] = (a.value, b.value)
out["mu_lambda"] = mulam
for atype in (2, 0):
    fw = np.zeros((len(lens), db.n), REFSW_DT)
    rv = np.zeros((len(lens), db.n), REFSW_DT)
    al = np.zeros((len(lens), db.n), REFALN_DT)
    cigs = []
    for qi in range(len(lens)):
        R.ref_structure_align(qa[qi], q3[qi], lens[qi], atype, 1, 0.5, 10, 1, db.dataaa, t3, db.offsets[:-1].copy(), db.lengths, db.n,
                              db.residues, 10.0, 0, 1, fw[qi].ctypes.data, rv[qi].ctypes.data, None, None, 0)
        buf = C.create_string_buffer(2_000_000)
        R.ref_structure_align(qa[qi], q3[qi], lens[qi], atype, 1, 0.5, 10, 1, db.dataaa, t3, db.offsets[:-1].copy(), db.lengths, db.n,
                              db.residues, 10.0, 1, 1, None, None, al[qi].ctypes.data, buf, 2_000_000)
        cigs.append(buf.value.decode())
    out[f"sw_fwd_t{atype}"] = fw
    out[f"sw_rev_t{atype}"] = rv
    out[f"aln_t{atype}"] = al
    out[f"cigars_t{atype}"] = np.array(cigs)
np.savez_compressed(os.path.join(HERE, "hotpath_v1.npz"), **out)
print("wrote", os.path.join(HERE, "hotpath_v1.npz"), os.path.getsize(os.path.join(HERE, "hotpath_v1.npz")), "bytes;",
      "accepted alignments:", int((out["aln_t2"]["status"] == 0).sum()), int((out["aln_t0"]["status"] == 0).sum()))
