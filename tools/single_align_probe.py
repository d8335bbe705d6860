"""configs[1]'s align half, one query at a time against 100k targets: the single-query entry (fshost_search_align: k_sw, one pair per wave, both directions) against the
batch entry with ONE query (fshost_search_align_batch: k_sw3, device-built images, both directions in one submission).  Wall ms per call and records compared."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from foldseek_amd import api, synth
q3, qa = synth.make_queries(8, seed=4242, lo=250, hi=450)
db = synth.make_db(100000, (q3, qa), seed=20260923, homologs_per_query=50)
ctx = api.Context(0); ctx.load_db(db)
for at in (0, 2):
    par = api.default_params(); par.alignmentType = at
    s = api.Search(ctx, par)
    hits = [s.prefilter(q)["id"] for q in q3]
    for rep in range(3):
        t1 = t2 = 0.0; same = True
        for i in range(len(q3)):
            t = time.perf_counter(); a = s.align(qa[i], q3[i], hits[i]); t1 += time.perf_counter() - t
            t = time.perf_counter(); b = s.align_batch([qa[i]], [q3[i]], [hits[i]])[0]; t2 += time.perf_counter() - t
            same = same and a.tobytes() == b.tobytes()
    print("alignment-type %d: single-query entry %.3f ms per call, batch entry with one query %.3f ms per call, records identical: %s" % (at, 1e3 * t1 / len(q3), 1e3 * t2 / len(q3), same))
    s.close()
