"""Parses foldseek_amd/data/fs_params.h (generated numbers) for the tests -> numpy arrays."""
import os, re
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load():
    txt = open(os.path.join(ROOT, "foldseek_amd", "data", "fs_params.h")).read()
    out = {}
    for name in ("MAT3DI", "BLOSUM62"):
        n = int(re.search(rf"FS_{name}_N = (\d+);", txt).group(1))
        lam = float(re.search(rf"FS_{name}_LAMBDA = ([^;]+);", txt).group(1))
        back = np.array([float(x) for x in re.search(rf"FS_{name}_BACK\[\d+\] = \{{([^}}]+)\}}", txt).group(1).split(",")])
        sc = re.search(rf"FS_{name}_SCORE\[[^\]]+\] = \{{([^}}]+)\}}", txt, re.S).group(1)
        score = np.array([float(x) for x in sc.replace("\n", " ").split(",") if x.strip()]).reshape(n, n)
        out[name] = dict(n=n, lam=lam, back=back, score=score)
    return out
