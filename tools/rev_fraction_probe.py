import sys, numpy as np
sys.path.insert(0,'/root/repo')
from foldseek_amd import api, synth
q3, qa = synth.make_queries(40, seed=1000, lo=250, hi=450)
hq = synth.make_queries(8, seed=1000, lo=250, hi=450)
db = synth.make_db(100000, hq, seed=20260923, homologs_per_query=50)
ctx = api.Context(0); ctx.load_db(db)
for atype in (0, 2):
    par = api.default_params(); par.alignmentType = atype
    s = api.Search(ctx, par)
    tot = 0; npairs = 0; acc = 0
    for i in range(40):
        hits = s.prefilter(q3[i])
        st0 = s.stats()[6]
        res = s.align(qa[i], q3[i], hits["id"])
        tot += s.stats()[6] - st0; npairs += len(hits); acc += len(res)
    print("atype", atype, "pairs", npairs, "needing rev", tot, "frac %.3f" % (tot / npairs), "accepted", acc)
    # homolog-rich queries (the 8 the DB was planted with)
    tot = 0; npairs = 0; acc = 0
    for i in range(8):
        hits = s.prefilter(hq[0][i])
        st0 = s.stats()[6]
        res = s.align(hq[1][i], hq[0][i], hits["id"])
        tot += s.stats()[6] - st0; npairs += len(hits); acc += len(res)
    print("   planted queries: pairs", npairs, "needing rev", tot, "frac %.3f" % (tot / npairs), "accepted", acc)
    s.close()
