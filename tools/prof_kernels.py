#!/usr/bin/env python3
"""Small driver for rocprofv3 runs: one resident synthetic DB, a few gapless scans + SW batches of one query.
usage: prof_kernels.py [--targets N] [--qlen L] [--reps K] [--atype 0|2]"""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from foldseek_amd import api, synth

ap = argparse.ArgumentParser()
ap.add_argument("--targets", type=int, default=100000)
ap.add_argument("--qlen", type=int, default=380)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--atype", type=int, default=0)
a = ap.parse_args()
rng = np.random.default_rng(7)
q3 = [rng.choice(20, size=a.qlen).astype(np.uint8)]
qa = [rng.choice(20, size=a.qlen).astype(np.uint8)]
db = synth.make_db(a.targets, (q3, qa), seed=20260923, homologs_per_query=50)
ctx = api.Context(0)
ctx.load_db(db)
par = api.default_params()
par.alignmentType = a.atype
s = api.Search(ctx, par)
for r in range(a.reps):
    t0 = time.perf_counter()
    hits = s.prefilter(q3[0])
    t1 = time.perf_counter()
    res = s.align(qa[0], q3[0], hits["id"])
    t2 = time.perf_counter()
    print(f"rep {r}: prefilter {1e3*(t1-t0):.3f} ms (kernel {ctx.kernel_ms(0):.3f}), align {1e3*(t2-t1):.3f} ms (kernel {ctx.kernel_ms(1):.3f}), hits {len(hits)}, accepted {len(res)}; host stages ms: " + " ".join(f"{1e3*x:.3f}" for x in s.stats()[:6]))
