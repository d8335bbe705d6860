#!/usr/bin/env python3
"""CPU-only sweep (TEST INFRASTRUCTURE): the block aligner restatement (foldseek_amd/csrc/host/block_aligner.cpp, through oracle/ba_kat/ba_kat.cpp) against
the independent trajectory model of the crate (tests/ba_model.py) in the CALL SHAPE OF alignStartPosBacktraceBlock (F/src/commons/StructureSmithWaterman.cpp:369-537:
reversed prefixes, both matrices, the query's composition bias, block sizes 32, 64, ... until the target score is reached, x-drop = -(size * extend + open)) on
freshly seeded homolog pairs -- longer than the committed cases (up to `maxlen` residues, so that block sizes beyond 64 and many shifts occur), four families
(plain, homopolymer stretch, tandem repeat, two-state low complexity), four gap-cost pairs.
usage: ba_model_sweep.py [seed=1] [cases=300] [maxlen=400] [dump=FILE] [per_boundary=40]
prints the number of compared cases / mismatches, the block sizes reached and the decision-boundary counts; with dump=FILE the inputs on which one of the RARE
decision boundaries fired (x-drop threshold met exactly / missed by one, second bad x-drop step, shrink with equality, a Grow that grew again) are appended to FILE
in the format of oracle/ba_kat/cases.txt (at most per_boundary per boundary kind) -- the cases to freeze for the day the harness runs against the Rust crate"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle", "ba_kat"))
from foldseek_amd import api, synth  # noqa: E402
from ba_model import BlockModel  # noqa: E402
import make_cases as MC  # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ncases = int(sys.argv[2]) if len(sys.argv) > 2 else 300
maxlen = int(sys.argv[3]) if len(sys.argv) > 3 else 400
dump = sys.argv[4] if len(sys.argv) > 4 else None
per_boundary = int(sys.argv[5]) if len(sys.argv) > 5 else 40
rng = np.random.default_rng(seed)
KAT = os.path.join(ROOT, "oracle", "ba_kat")
w = tempfile.mkdtemp(prefix="ba_sweep_")
m3, mA = api.Matrix(0, 2.1, 0.0), api.Matrix(1, 1.4, 0.0)
MC.write_matrix(os.path.join(w, "mat_3di.txt"), m3); MC.write_matrix(os.path.join(w, "mat_aa.txt"), mA)
s3, sA = m3.scores().astype(np.int64), mA.scores().astype(np.int64)
L = MC.LETTERS
lines, trial = ["# ba_model_sweep"], 0
while len(lines) - 1 < ncases:
    trial += 1
    fam = trial % 4
    Lq = int(rng.integers(32, maxlen))
    q3 = rng.choice(20, size=Lq, p=synth.BACK_3DI / synth.BACK_3DI.sum()).astype(np.uint8)
    qa = rng.choice(20, size=Lq, p=synth.BACK_AA / synth.BACK_AA.sum()).astype(np.uint8)
    if fam == 1:
        a = int(rng.integers(0, Lq - 12)); n = int(rng.integers(6, 30)); q3[a:a + n] = q3[a]; qa[a:a + n] = qa[a]
    elif fam == 2:
        u = int(rng.integers(2, 6)); a = int(rng.integers(0, max(1, Lq - 8 * u)))
        for k in range(min(8 * u, Lq - a)):
            q3[a + k] = q3[a + k % u]; qa[a + k] = qa[a + k % u]
    elif fam == 3:
        q3 = rng.choice(q3[:2], size=Lq).astype(np.uint8); qa = rng.choice(qa[:3], size=Lq).astype(np.uint8)
    t3, ta = synth._mutate(rng, q3, qa, float(rng.choice([0.1, 0.2, 0.35])), float(rng.choice([0.03, 0.10, 0.2])))
    if rng.random() < 0.4:          # a long insertion / deletion: the x-drop run of a small block gives up, the caller retries with the next size
        n = int(rng.integers(20, 160)); a = int(rng.integers(1, max(2, len(t3) - 1)))
        if rng.random() < 0.5:
            t3 = np.concatenate([t3[:a], rng.integers(0, 20, n).astype(np.uint8), t3[a:]]); ta = np.concatenate([ta[:a], rng.integers(0, 20, n).astype(np.uint8), ta[a:]])
        elif len(t3) > n + 40:
            t3 = np.concatenate([t3[:a], t3[a + n:]]); ta = np.concatenate([ta[:a], ta[a + n:]])
    pad = int(rng.integers(0, 40))
    t3 = np.concatenate([rng.integers(0, 20, pad).astype(np.uint8), t3]); ta = np.concatenate([rng.integers(0, 20, pad).astype(np.uint8), ta])
    if len(t3) < 10:
        continue
    _, _, cbA, cbS = api.align_profiles(mA, m3, qa, q3, comp_bias=True, scale=0.5)
    bias = cbA.astype(np.int64) + cbS.astype(np.int64)
    go, ge = [(10, 1), (10, 1), (8, 2), (3, 1), (15, 3)][int(rng.integers(0, 5))]
    S = s3[q3][:, t3] + sA[qa][:, ta] + bias[:, None]
    best, qe, te = MC.best_local_end(S, go, ge)
    if best < 25:
        continue
    rev = lambda x, e: "".join(L[c] for c in x[:e + 1][::-1])  # noqa: E731
    qb = ",".join(str(int(b)) for b in bias[:qe + 1][::-1])
    lines.append(f"3di fam{fam}_{trial}@{best} {go} {ge} {rev(qa, qe)} {rev(q3, qe)} {qb} {rev(ta, te)} {rev(t3, te)}")
open(os.path.join(w, "cases.txt"), "w").write("\n".join(lines) + "\n")
exe = os.path.join(w, "ba_kat_ours")
subprocess.check_call(["g++", "-O2", "-std=c++17", "-mavx2", "-mfma", "-I" + os.path.join(ROOT, "foldseek_amd", "csrc", "host"), "-o", exe,
                       os.path.join(KAT, "ba_kat.cpp"), os.path.join(ROOT, "foldseek_amd", "csrc", "host", "block_aligner.cpp")])
ours = subprocess.run([exe, w], stdout=subprocess.PIPE, text=True, check=True).stdout.splitlines()


def load(path):
    toks = open(path).read().split()
    letters, vals = toks[0], list(map(int, toks[1:]))
    n = len(letters)
    tab = {}
    for a in range(n):
        for b in range(n):
            tab[(letters[a], letters[b])] = vals[a * n + b]
            tab[(letters[b], letters[a])] = vals[a * n + b]
    return lambda x, y: tab.get((x, y), 1 if x == y else -1)


fA, f3 = load(os.path.join(w, "mat_aa.txt")), load(os.path.join(w, "mat_3di.txt"))
bad, sizes_seen, ties, missed = 0, {}, {}, 0
RARE = ("xdrop_at_threshold", "xdrop_one_below", "xdrop_second_step", "shrink_equal", "grow_twice")
dumped, dump_lines = {k: 0 for k in RARE}, []
for k, line in enumerate(lines[1:]):
    f = line.split()
    name, go, ge = f[1], int(f[2]), int(f[3])
    qa, q3, qb, ta, t3 = f[4:9]
    target = int(name.split("@")[1])
    qbias = ([int(x) for x in qb.split(",")] + [0] * len(qa))[:len(qa)]
    res, sizes, ms = (-10 ** 9, 0, 0), [], 32
    fired = set()
    while ms <= 4096 and res[0] < target:
        M = BlockModel(qa, ta, fA, -go, -ge, ms, 4096, x_drop=-(ms * (-ge) + (-go)), q_bias=qbias, r_bias=[0] * len(ta), score2=f3, q2=q3, r2=t3)
        res = M.align()
        sizes.append(f"{ms}:{res[0]}")
        for kk, v in M.ties.items():
            ties[kk] = ties.get(kk, 0) + (1 if v else 0)
            if v:
                fired.add(kk)
        ms *= 2
    got = "\t".join([name, str(res[0]), str(res[1]), str(res[2]), M.trace.cigar(res[1], res[2]) or "-", ",".join(sizes)])
    sizes_seen[len(sizes)] = sizes_seen.get(len(sizes), 0) + 1
    missed += res[0] != target
    want = [kk for kk in RARE if kk in fired and dumped[kk] < per_boundary]
    if dump and want:
        for kk in want:
            dumped[kk] += 1
        f2 = line.split()
        f2[1] = "swp" + str(seed) + "_" + "+".join(w_[:8] for w_ in want).replace("xdrop_", "xd") + "_" + f2[1]
        dump_lines.append(" ".join(f2))
    if got != ours[k]:
        bad += 1
        print("MISMATCH", k, "\n  model", got[:300], "\n  ours ", ours[k][:300])
print(f"seed {seed}: {len(lines) - 1} cases up to {maxlen} residues, {bad} mismatches, target score missed {missed}x, block-size attempts per case {dict(sorted(sizes_seen.items()))}, decision boundaries {ties}")
if dump:
    with open(dump, "a") as fh:
        fh.write("".join(x + "\n" for x in dump_lines))
    print(f"appended {len(dump_lines)} boundary cases to {dump}: {dumped}")
