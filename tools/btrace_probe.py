"""Latency / throughput of the device block-aligner backtrace against the host pool: one align_batch call over NQ queries (their prefilter hit lists with planted
homologs) with FSGPU_DEVICE_BACKTRACE = 1 and 0.  usage: btrace_probe.py [targets=100000] [queries=1,32,256]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from foldseek_amd import api, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
NQS = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,32,256").split(",")]
q3, qa = synth.make_queries(max(NQS), seed=4100, lo=250, hi=450)
db = synth.make_db_fast(N, (q3, qa), seed=20260924, homologs_per_query=50)
ctx = api.Context(0); ctx.load_db(db)
par = api.default_params(); par.addBacktrace = 1
pre = api.Search(ctx)
hits = pre.prefilter_batch(q3)
s = api.Search(ctx, par)
for nq in NQS:
    for dev in ("1", "0", "1", "0"):
        os.environ["FSGPU_DEVICE_BACKTRACE"] = dev
        t = time.perf_counter()
        res = s.align_batch(qa[:nq], q3[:nq], [h["id"] for h in hits[:nq]])
        dt = time.perf_counter() - t
        st = s.stats()
        print(f"queries {nq:4d} device {dev}: align_batch {1e3 * dt:8.2f} ms, backtrace part {1e3 * st[5]:8.2f} ms, accepted {sum(len(r) for r in res)}, on device {s.backtrace_counts()}", flush=True)
