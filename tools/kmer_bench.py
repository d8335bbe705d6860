"""k-mer prefilter throughput on the GPU: NQ queries against a synthetic N-target DB, batches of 32 through the C ABI."""
import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from foldseek_amd import api, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
NQ = int(sys.argv[2]) if len(sys.argv) > 2 else 128
REPS = int(sys.argv[3]) if len(sys.argv) > 3 else 3
BATCH = int(sys.argv[4]) if len(sys.argv) > 4 else 32          # queries per fsgpu_kmer_search call
q3, qa = synth.make_queries(NQ, seed=1)
db = synth.make_db(N, (q3[:32], qa[:32]))
ctx = api.Context(0); ctx.load_db(db)
m8 = api.Matrix(0, 8.0, -0.2); m2 = api.Matrix(0, 2.0, -0.2)
t = time.time(); ctx.kmer_index_build(m8, kmer_thr=78); print("index build %.3fs entries=%d" % (time.time() - t, ctx.kmer_index_entries), flush=True)
t = time.time(); prep = [api.kmer_query_prepare(m8, m2, q) for q in q3]; print("host prepare %.3f ms/query" % ((time.time() - t) / NQ * 1e3))
for rep in range(REPS):
    stages = np.zeros(11); t = time.time(); hits = 0; counts = np.zeros(4)
    for b in range(0, NQ, BATCH):
        res, status, stats = ctx.kmer_search(prep[b:b + BATCH], max_res=1000, want_stats=True)
        stages += np.array(ctx.kmer_stage_ms()); hits += stats[:, 1].sum(); counts += np.array(ctx.kmer_counts(), dtype=np.float64)
        assert (status >= 0).all()
    dt = time.time() - t
    print("partition of the last batch [-, runs, -, tiles, runs, coarse keys, ids of the widest key]:", ctx.kmer_segments() if hasattr(ctx, "kmer_segments") else None)
    print("rep %d: %.3f ms/query wall, device %.3f ms/query; stage ms/query %s; hits/query %.0f; prefilter residues/s %.3e" % (
        rep, dt / NQ * 1e3, stages[0] / NQ, ["%.3f" % (x / NQ) for x in stages[1:]], hits / NQ, NQ * db.residues / dt), flush=True)
    print("COUNTS " + __import__("json").dumps({"targets": N, "queries": NQ, "similar_kmers": counts[0], "index_hits": counts[1], "candidates": counts[2],
                                                   "mean_query_len": float(np.mean([len(q) for q in q3])), "device_ms": stages[0]}), flush=True)
