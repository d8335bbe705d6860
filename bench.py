#!/usr/bin/env python3
"""bench.py -- residues aligned / s of the hot path (prefilter + structurealign) on N MI355X of one node.

Workload = BASELINE.json's metric configuration (configs[2], it fits one GPU): queries vs a 1M-structure synthetic
3Di(+AA) database (mean length 350, 50 planted homologs for EVERY query) resident in HBM.
One STEP = one batch of --group (64) queries through the whole path on one host feeder thread: exhaustive gapless
prefilter over every target + top --max-seqs selection per query, then ONE multi-query structure Smith-Waterman launch
per pass over the batch's hit lists (forward over all pairs, reversed query over the pairs that pass the forward
gates), host gates, block-aligner backtrace of every accepted hit, result ordering.  --host-threads feeder threads
(own HIP stream each, shared resident DB) run their steps concurrently, the way the reference's OpenMP threads do.
--targets 100000 gives configs[1].

N > 1: `python bench.py --gpus N` starts N ranks itself (torch.distributed.run, one rank per GPU; under torchrun it
uses the ranks it is given): rank 0 generates the padded DB, ONE RCCL broadcast puts it into every GPU's HBM, every rank
then searches its own queries with no further communication.  --scaling weak (default): every rank runs --steps
steps of its own queries; --scaling strong: the SAME steps x group queries are split over the ranks.

One JSON line on rank 0; see the prompt contract.  `roofline` is for the dominant kernel (gapless scan) from HIP
events around that kernel on the library's stream; `cpu_baseline` times the reference's own AVX2 code (oracle/_ref,
kind "reference") or, if that was not built, the C port (oracle/libfso.so) on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1, help="ranks = GPUs of this node; > 1 without a torchrun environment: bench.py launches them itself")
    ap.add_argument("--steps", type=int, default=20, help="timed steps per rank (weak) / in total (strong); a step = one batch of --group queries")
    ap.add_argument("--warmup", type=int, default=5, help="untimed steps per rank")
    ap.add_argument("--targets", type=int, default=1000000)
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--homologs", type=int, default=50, help="planted homologs per query (SURVEY.md 8d)")
    ap.add_argument("--query-len", default="250,450", help="lo,hi: query lengths are drawn from the DB's length model (mean 350) clipped to this range; "
                    "30,2000 = the DB's own range (queries longer than 512 residues run row-tiled)")
    ap.add_argument("--alignment-type", type=int, default=0, help="0: 3Di only (configs[1..2]), 2: 3Di+AA (configs[3])")
    ap.add_argument("--host-threads", type=int, default=3, help="host feeder threads per GPU (each with its own stream)")
    ap.add_argument("--group", type=int, default=64, help="queries per step: prefiltered back to back, then ONE multi-query SW launch per pass")
    ap.add_argument("--workload", choices=["search", "allvsall"], default="search",
                    help="search: queries vs the DB (configs[1..3]); allvsall: every DB entry searches the DB (configs[4], easy-cluster's "
                         "cascaded step: k-mer prefilter -s 4.5 --max-seqs 200 + structurealign -e 0.01 -c 0.8), use with --targets 200000")
    ap.add_argument("--dry-run", action="store_true", help="no device work: ranks, DB generation, broadcast and query sharding only (gloo on CPU when no GPU is visible)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kmer", action="store_true", help="skip the k-mer prefilter (+align) section")
    ap.add_argument("--kmer-threads", type=int, default=4, help="host threads (context clones) of the k-mer section")
    ap.add_argument("--kmer-queries", type=int, default=512, help="queries of the k-mer prefilter section (batches of 32)")
    ap.add_argument("--kmer-cpu-queries", type=int, default=64, help="queries the reference k-mer prefilter is timed on")
    ap.add_argument("--cpu-sample-targets", type=int, default=100000)
    ap.add_argument("--cpu-sample-queries", type=int, default=16)
    return ap.parse_args(argv)


def relaunch_as_ranks(args):
    """`python bench.py --gpus N` outside torchrun: start N ranks of this script on this node (one per GPU) the way the
    driver would (torch.distributed.run, rendezvous on 127.0.0.1) and pass its exit code on."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def usable_cores():
    """CPUs this process may actually use: the affinity mask capped by the cgroup CPU quota (a container that shows 256
    logical CPUs can be limited to the time of 16; running 256 OpenMP threads there only measures the throttling)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, quota // period))
        except (OSError, ValueError):
            pass
    return max(1, n)


def cpu_baseline(db, q3, qa, hits_ids, atype, sample_targets, more_queries=()):
    """reference AVX2 code (or the C port) on the host cores: prefilter over a target sample + align over the hit list.
    The prefilter is timed over several queries back to back (about 20 core-seconds of work at the defaults) so that the
    OpenMP start-up of a single 5 ms parallel region does not dominate the figure."""
    import oracle_lib
    threads = usable_cores()
    ref = oracle_lib.load_ref()
    ns = min(sample_targets, db.n)
    # spread the sample over the length-sorted DB so its residue mix matches the full DB
    idx = np.linspace(0, db.n - 1, ns).astype(np.int64)
    offs = np.ascontiguousarray(db.offsets[:-1][idx])
    lens = np.ascontiguousarray(db.lengths[idx])
    sample_res = int(lens.sum())
    if ref is not None:
        scores = np.zeros(ns, np.int32)
        qs = [q3] + [q for q in more_queries]
        ref.ref_ungapped(q3, len(q3), 1, 0.15, db.data3di, offs, lens, ns, threads, scores)          # warm the thread pool
        # best of 2 passes over the query set: the reference's "omp for schedule(static)" over targets is noisy at this thread count
        t_pref = min(sum(ref.ref_ungapped(q, len(q), 1, 0.15, db.data3di, offs, lens, ns, threads, scores) for q in qs) for _ in range(2)) / len(qs)
        t3 = np.where(db.data3di >= 32, db.data3di - 32, db.data3di).astype(np.uint8)
        h = np.ascontiguousarray(hits_ids.astype(np.int64))
        aln = np.zeros(max(1, len(h)), oracle_lib.REFALN_DT)
        t_aln = min(ref.ref_structure_align(qa, q3, len(q3), atype, 1, 0.5, 10, 1, db.dataaa, t3,
                                            np.ascontiguousarray(db.offsets[:-1][h]), np.ascontiguousarray(db.lengths[h]), len(h),
                                            db.residues, 10.0, 1, threads, None, None, aln.ctypes.data, None, 0) for _ in range(3)) if len(h) else 0.0
        kind = "reference"
    else:
        import helpers
        threads = 1
        t0 = time.perf_counter()
        sub = synth_sub = None
        O = helpers.oracle()
        sub, pb = helpers.o_submat("MAT3DI", 2.0)
        cb = helpers.o_round_bias(sub, pb, q3, 0.15)[1]
        tiny = sub.astype(np.int8)
        ns = min(ns, 2000)
        idx = idx[:: max(1, len(idx) // ns)][:ns]
        sample_res = int(db.lengths[idx].sum())
        for i in idx:
            raw = db.data3di[db.offsets[i]:db.offsets[i] + db.lengths[i]]
            O.fso_ungapped_score(q3, len(q3), tiny, 21, cb, np.ascontiguousarray(np.where(raw >= 32, 20, raw).astype(np.uint8)), int(db.lengths[i]))
        t_pref = time.perf_counter() - t0
        t_aln = 0.0
        kind = "port"
    t_full = t_pref * (db.residues / max(1, sample_res)) + t_aln
    nqs = 1 + len(more_queries) if ref is not None else 1
    return {"value": db.residues / t_full, "unit": "residues/s", "cores": threads, "kind": kind,
            "sample": f"gapless prefilter timed on {ns} of {db.n} targets ({sample_res} residues, scaled linearly), mean over {nqs} queries run back to back + "
                      f"fwd/rev structure SW on the {len(hits_ids)} prefilter hits of one query, {threads} host threads "
                      f"(= usable cores: affinity mask capped by the cgroup CPU quota; {os.cpu_count()} logical CPUs visible)",
            "prefilter_s_sample": t_pref, "align_s": t_aln}


def kmer_section(args, api, synth, ctx0, search0, par, db, rank, world, dev, fdist, q3, qa):
    """k-mer prefilter (Foldseek's default prefilter on CPUs) + structure SW on its hits, same resident DB.
    Every rank builds its own index from the broadcast DB (no collective) and searches its own queries."""
    import threading
    nqk = max(32, args.kmer_queries // 32 * 32)
    # this rank's own timed queries (their homologs are planted in the DB), repeated if the section asks for more
    q3 = [q3[i % len(q3)] for i in range(nqk)]
    qa = [qa[i % len(qa)] for i in range(nqk)]
    m8, m2 = api.Matrix(0, 8.0, -0.2), api.Matrix(0, 2.0, -0.2)
    thr = api.kmer_threshold(9.5, 6)
    t0 = time.perf_counter()
    ctx0.kmer_index_build(m8, kmer_thr=thr)
    t_index = time.perf_counter() - t0
    # a clone made now shares the resident DB and the index; two host threads keep the device busy during the host tails
    KT = max(1, args.kmer_threads)
    kctx = [ctx0] + [ctx0.clone() for _ in range(KT - 1)]
    ksearch = [search0] + [api.Search(c, par) for c in kctx[1:]]
    prep = [api.kmer_query_prepare(m8, m2, q, kmer_thr=thr) for q in q3]
    batches = [list(range(b, b + 32)) for b in range(0, nqk, 32)]
    stat = {"dev": [], "lists": [], "counts": [], "hits": 0, "aln": 0, "t_pref": 0.0, "t_aln": 0.0, "hq": 0}
    lock = threading.Lock()

    def run(t, ids, timed):
        tp = time.perf_counter()
        res, status = kctx[t].kmer_search([prep[i] for i in ids], max_res=1000)
        tp = time.perf_counter() - tp
        ms, cnt = kctx[t].kmer_stage_ms(), kctx[t].kmer_counts()
        ta = time.perf_counter()
        aln = ksearch[t].align_batch([qa[i] for i in ids], [q3[i] for i in ids], [r["id"] for r in res])
        na = sum(len(a) for a in aln)
        ta = time.perf_counter() - ta
        swms = kctx[t].kernel_ms(1)
        if timed:
            with lock:
                stat["dev"].append(ms[0]); stat["lists"].append(ms[10]); stat["counts"].append(cnt); stat.setdefault("sw", []).append(swms)
                stat["hits"] += sum(len(r) for r in res); stat["aln"] += na; stat["t_pref"] += tp; stat["t_aln"] += ta
                stat["bad"] = stat.get("bad", 0) + int((status < 0).sum())
        return res

    for t in range(KT):
        run(t, batches[0], False)                                 # warm every host thread's context
    ready, go = threading.Barrier(KT + 1), threading.Barrier(KT + 1)

    def worker(t):
        ready.wait(); go.wait()
        for b in range(t, len(batches), KT):
            run(t, batches[b], True)

    ths = [threading.Thread(target=worker, args=(t,)) for t in range(KT)]
    for th in ths:
        th.start()
    ready.wait()
    import torch
    torch.cuda.synchronize()
    import gc
    gc.collect()
    gc.disable()            # a generation-2 collection of the interpreter (torch is imported: ~50 ms) would stall every feeder thread at once
    t0 = time.perf_counter()
    go.wait()
    for th in ths:
        th.join()
    torch.cuda.synchronize()
    dt = fdist.max_over_ranks(time.perf_counter() - t0, dev)
    gc.enable()
    # solo batch on an idle GPU for the roofline of the dominant kernel
    run(0, batches[-1], False)
    solo_ms, solo_cnt = kctx[0].kmer_stage_ms(), kctx[0].kmer_counts()
    for x in ksearch[1:]:
        x.close()
    for c in kctx[1:]:
        c.close()
    if rank != 0:
        return None
    probes = float(solo_cnt[0])
    lists_s = solo_ms[10] * 1e-3
    alg = probes * 8.0                                         # two uint32 offsets per similar k-mer
    reg_ms = float(np.mean(stat["lists"]))
    reg_probes = float(np.mean([c[0] for c in stat["counts"]]))
    reg_alg = reg_probes * 8.0
    traffic, traffic_src = None, None
    try:   # HBM bytes per probe from a committed rocprofv3 --pmc pass of the same kernel on the same DB size
        e = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic_kmer.json"))).get(str(args.targets))
        if e:
            traffic = e["k_kmer_lists_bytes_per_probe"] * probes
            traffic_src = e["source"]
    except Exception:
        traffic = None
    out = {"workload": f"{nqk} queries in batches of 32 vs the same {db.n}-structure DB: k-mer prefilter (-s 9.5, k=6 spaced, "
                       f"--max-seqs 1000, double-diagonal + ungapped scoring) + fwd/rev structure SW on its hits (one multi-query SW launch per register class)",
           "value": world * nqk * db.residues / dt, "unit": "residues/s", "queries_per_s": world * nqk / dt,
           "ms_per_query": 1e3 * dt / nqk, "prefilter_ms_per_query_host_wall": 1e3 * stat["t_pref"] / nqk,
           "align_ms_per_query_host_wall": 1e3 * stat["t_aln"] / nqk, "sw_kernels_ms_per_batch32": float(np.mean(stat["sw"])), "prefilter_device_ms_per_query": float(np.sum(stat["dev"])) / nqk,
           "host_threads": KT, "index_build_s": t_index, "index_entries": int(ctx0.kmer_index_entries), "kmer_threshold": thr,
           "similar_kmers_per_query": float(np.mean([c[0] for c in stat["counts"]])) / 32, "index_hits_per_query": float(np.mean([c[1] for c in stat["counts"]])) / 32,
           "candidates_per_query": float(np.mean([c[2] for c in stat["counts"]])) / 32,
           "hits_per_query": stat["hits"] / nqk, "alignments_per_query": stat["aln"] / nqk, "unsupported_queries": stat.get("bad", 0),
           "stage_ms_per_batch32_solo": {k: solo_ms[i] for i, k in enumerate(["device_total", "count", "lists", "emit", "sort", "dupflags", "score", "replay", "select", "host_tail", "k_kmer_lists"])},
           # k_kmer_lists per-launch duration: HIP events on the library's stream, mean over the batches of the timed region
           # (the host threads overlap their batches); "solo" = the same launch alone on the device
           "roofline": {"bound": "hbm", "kernel": "k_kmer_lists", "kernel_ms": reg_ms, "achieved": reg_alg / (reg_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                        "frac": reg_alg / (reg_ms * 1e-3) / 1e9 / 8000.0, "traffic": None if traffic is None else traffic * reg_probes / probes, "traffic_source": traffic_src,
                        "algorithmic_bytes": reg_alg, "probes_per_launch": reg_probes, "probes_per_s": reg_probes / (reg_ms * 1e-3),
                        "solo": {"kernel_ms": solo_ms[10], "achieved": alg / lists_s / 1e9, "frac": alg / lists_s / 1e9 / 8000.0, "probes_per_launch": probes,
                                 "traffic": traffic},
                        "note": "random 8-byte probes of the 256 MB k-mer offset table (algorithmic bytes = 8 per similar k-mer); the hardware "
                                "moves a sector per probe of a non-empty list, see DESIGN.md 4.5"}}
    return out


def kmer_cpu_baseline(args, synth, db):
    """the reference's own k-mer prefilter classes (oracle/_ref) on all host cores: index build + matchQuery"""
    import kmer_lib as K
    R = K.load_ref()
    if R is None:
        return None
    threads = usable_cores()
    nq = args.kmer_cpu_queries
    q3, _ = synth.make_queries(nq, seed=1000, lo=250, hi=450)       # the first queries of the timed set
    t0 = time.perf_counter()
    r = K.RefKpf(R, [db.seq(i, "3di", unmask=False) for i in range(db.n)], threads=threads)
    t_build = time.perf_counter() - t0
    _, _, secs = r.run(q3, None, threads=threads)
    _, _, secs = r.run(q3, None, threads=threads)
    r.close()
    return {"value": nq * db.residues / secs, "unit": "residues/s", "queries_per_s": nq / secs, "cores": threads, "kind": "reference",
            "sample": f"QueryMatcher::matchQuery of the reference on {nq} queries vs the same {db.n}-target DB, {threads} OpenMP threads "
                      f"(dynamic,1), second of two runs; index build {t_build:.1f}s not included; prefilter only (no alignment)",
            "index_build_s": t_build}


def allvsall(args, api, synth, fdist, dev, rank, world, local_rank):
    """configs[4]: all-vs-all of a --targets structure DB, the shape of easy-cluster's cascaded steps (F/data/structurecluster.sh via
    `easy-cluster -v 3`): prefilter -s 4.5 --max-seqs 200 --min-ungapped-score 30 -c 0.8 --add-self-matches 1, then structurealign
    -e 0.01 -c 0.8 --comp-bias-corr 0.  Prefilter dominated: one pass of the k-mer index per batch of 32 queries.  Every DB entry is a
    query; the ids shard over the ranks (--steps batches each, 0 = all of them), the DB is replicated by one broadcast."""
    import threading
    import torch
    import torch.distributed as dist
    t_gen = time.perf_counter()
    db = synth.make_db_fast(args.targets, None, seed=20260923, homologs_per_query=0) if rank == 0 else None
    t_gen = time.perf_counter() - t_gen
    tensors, db = fdist.broadcast_db(db, dev)
    torch.cuda.synchronize()
    ctx0 = api.Context(local_rank)
    ctx0.adopt_device_db(tensors[0].data_ptr(), tensors[1].data_ptr(), tensors[2].data_ptr(), tensors[3].data_ptr(), db.n, db.data3di.size)
    ctx0._keep = (np.ascontiguousarray(db.data3di), np.ascontiguousarray(db.dataaa), np.ascontiguousarray(db.offsets, np.uint64), np.ascontiguousarray(db.lengths, np.int32))
    del tensors
    par = api.default_params()
    par.alignmentType = 2
    par.evalThr, par.covThr, par.covMode, par.compBiasCorrection = 0.01, 0.8, 0, 0
    m8, m2 = api.Matrix(0, 8.0, -0.2), api.Matrix(0, 2.0, -0.2)
    thr = api.kmer_threshold(4.5, 6)
    t0 = time.perf_counter()
    ctx0.kmer_index_build(m8, kmer_thr=thr)
    t_index = time.perf_counter() - t0
    KT = max(1, args.kmer_threads)
    ctxs = [ctx0] + [ctx0.clone() for _ in range(KT - 1)]
    searches = [api.Search(c, par) for c in ctxs]
    lo, hi = fdist.shard_range(db.n, rank, world)
    ids = np.arange(lo, hi)
    nb_all = (len(ids) + 31) // 32
    nb = nb_all if args.steps <= 0 else min(nb_all, args.steps + args.warmup)
    # a spread sample of the shard when only some batches are run: the DB is length sorted
    pick = np.linspace(0, nb_all - 1, nb).astype(np.int64) if nb < nb_all else np.arange(nb_all)
    batches = [ids[b * 32:(b + 1) * 32] for b in pick]
    warm, timed = (batches[:args.warmup], batches[args.warmup:]) if args.steps > 0 else (batches[:1], batches)
    stat = {"hits": 0, "aln": 0, "q": 0, "res": 0, "dev": 0.0, "bad": 0}
    lock = threading.Lock()

    def run(t, b, count):
        q3 = [db.seq(int(i), "3di") for i in b]
        qa = [db.seq(int(i), "aa") for i in b]
        prep = [api.kmer_query_prepare(m8, m2, q, kmer_thr=thr) for q in q3]
        res, status = ctxs[t].kmer_search(prep, identity=b, max_res=200)
        # Prefiltering.cpp:880-887: canBeCovered at -c 0.8, cov-mode 0
        keep = []
        for q, r in zip(q3, res):
            lt = db.lengths[r["id"]].astype(np.float32)
            lq = np.float32(len(q))
            keep.append(r["id"][(lq / lt >= 0.8) & (lt / lq >= 0.8)])
        aln = searches[t].align_batch(qa, q3, keep, identity=b)
        if count:
            with lock:
                stat["hits"] += sum(len(k) for k in keep); stat["aln"] += sum(len(a) for a in aln); stat["q"] += len(b)
                stat["res"] += int(sum(len(q) for q in q3)); stat["dev"] += ctxs[t].kmer_stage_ms()[0]; stat["bad"] += int((status < 0).sum())

    for t in range(KT):
        for b in warm[t::KT] or warm[:1]:
            run(t, b, False)
    ready, go = threading.Barrier(KT + 1), threading.Barrier(KT + 1)

    def worker(t):
        ready.wait(); go.wait()
        for b in timed[t::KT]:
            run(t, b, True)

    ths = [threading.Thread(target=worker, args=(t,)) for t in range(KT)]
    for th in ths:
        th.start()
    ready.wait()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    import gc
    gc.collect(); gc.disable()
    t0 = time.perf_counter()
    go.wait()
    for th in ths:
        th.join()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = fdist.max_over_ranks(time.perf_counter() - t0, dev)
    gc.enable()
    tot = fdist.gather_objects(stat)
    if rank == 0:
        nq = sum(x["q"] for x in tot)
        out = {"metric": "residues aligned/sec (prefilter+align)", "value": nq * db.residues / dt, "unit": "residues/s", "n_gpus": world,
               "steps": len(timed), "warmup": len(warm), "ms_per_step": 1e3 * dt / max(1, len(timed)), "higher_is_better": True,
               "scaling": "strong" if args.steps <= 0 else "weak", "vs_baseline": None, "dtype": "u8 k-mer index probes / diagonal scores + i16 SW", "data": "synthetic",
               "config": {"workload": f"all-vs-all (configs[4]): {nq} of the {db.n} DB entries as queries in batches of 32 (1 step = 1 batch), k-mer prefilter -s 4.5 --max-seqs 200 "
                                      f"-c 0.8 + structurealign -e 0.01 -c 0.8 (3Di+AA) on its hits; queries shard over {world} rank(s), DB replicated by one broadcast",
                          "targets": db.n, "db_residues": db.residues, "host_threads_per_gpu": KT, "kmer_threshold": thr},
               "queries_per_s": nq / dt, "ms_per_query": 1e3 * dt / max(1, nq / world), "hits_per_query": sum(x["hits"] for x in tot) / max(1, nq),
               "alignments_per_query": sum(x["aln"] for x in tot) / max(1, nq), "prefilter_device_ms_per_query": tot[0]["dev"] / max(1, tot[0]["q"]),
               "unsupported_queries": sum(x["bad"] for x in tot), "index_build_s": t_index, "db_generation_s": t_gen,
               "projected_full_all_vs_all_s": db.n / max(1e-9, nq / dt)}
        print(json.dumps(out))
    for x in searches:
        x.close()
    for c in ctxs[1:]:
        c.close()
    ctx0.close()
    if world > 1:
        dist.destroy_process_group()


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(relaunch_as_ranks(args))
    # host feeder threads spend their time inside the library (ctypes releases the GIL); a thread returning from a call
    # must not wait a whole default switch interval (5 ms) for the GIL while another one runs a few Python lines
    sys.setswitchinterval(1e-4)
    import torch
    import torch.distributed as dist
    from foldseek_amd import synth
    from foldseek_amd import dist as fdist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    have_gpu = torch.cuda.is_available()
    if not have_gpu and not args.dry_run:
        raise SystemExit("bench.py: no GPU visible (the hot path has no CPU fallback); --dry-run exercises the rank / broadcast / sharding plumbing only")
    # one rank per GPU.  FSGPU_BENCH_BACKEND=gloo + more ranks than devices (ranks share devices round robin) is a plumbing check
    # for boxes with a single GPU: everything but the RCCL transport of the one broadcast is the code the 8-GPU run executes.
    backend = os.environ.get("FSGPU_BENCH_BACKEND", "nccl" if have_gpu else "gloo")
    dev_index = local_rank % max(1, torch.cuda.device_count()) if have_gpu else 0
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
    if have_gpu:
        torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index) if have_gpu else torch.device("cpu")
    local_rank = dev_index

    if args.workload == "allvsall" and not args.dry_run:
        from foldseek_amd import api
        return allvsall(args, api, synth, fdist, dev, rank, world, local_rank)
    import threading
    nthreads = max(1, args.host_threads)
    G = max(1, args.group)
    # ---- queries: weak = every rank its own steps; strong = the same steps*G queries split contiguously over the ranks ----
    per_rank_timed = args.steps * G if args.scaling == "weak" else None
    n_timed_total = args.steps * G * (world if args.scaling == "weak" else 1)
    n_warm = args.warmup * G
    q_lo, q_hi = (int(x) for x in args.query_len.split(","))
    all_q3, all_qa = synth.make_queries(n_timed_total + world * n_warm, seed=1000, lo=q_lo, hi=q_hi)   # default 250..450: around the mean length 350; same on every rank
    if args.scaling == "weak":
        t_lo, t_hi = rank * per_rank_timed, (rank + 1) * per_rank_timed
    else:
        t_lo, t_hi = fdist.shard_range(n_timed_total, rank, world)
    w_lo = n_timed_total + rank * n_warm
    q3 = all_q3[w_lo:w_lo + n_warm] + all_q3[t_lo:t_hi]            # this rank: warm-up queries first, then its timed shard
    qa = all_qa[w_lo:w_lo + n_warm] + all_qa[t_lo:t_hi]
    nq = len(q3)
    n_mine = t_hi - t_lo
    # ---- target DB: generated on rank 0 (vectorised), ONE broadcast (RCCL over xGMI), then resident in every GPU's HBM ----
    t_gen = time.perf_counter()
    db = synth.make_db_fast(args.targets, (all_q3, all_qa), seed=20260923, homologs_per_query=args.homologs) if rank == 0 else None
    t_gen = time.perf_counter() - t_gen
    if have_gpu:
        torch.cuda.synchronize()
    tb = time.perf_counter()
    tensors, db = fdist.broadcast_db(db, dev)
    if have_gpu:
        torch.cuda.synchronize()
    t_bcast = time.perf_counter() - tb if world > 1 else 0.0
    if args.dry_run:
        # everything up to here is the multi-rank plumbing; check what the other ranks received and stop
        import zlib
        digest = zlib.crc32(db.data3di.tobytes()[:1 << 20]) ^ zlib.crc32(np.ascontiguousarray(db.lengths).tobytes())
        sizes = fdist.gather_objects((rank, n_mine, int(db.n), int(digest)))
        if world > 1:
            dist.barrier()
        if rank == 0:
            print(json.dumps({"metric": "residues aligned/sec (prefilter+align)", "value": 0.0, "unit": "residues/s", "n_gpus": world,
                              "steps": args.steps, "warmup": args.warmup, "dry_run": True, "scaling": args.scaling,
                              "backend": dist.get_backend() if world > 1 else "none", "db_broadcast_s": t_bcast, "db_generation_s": t_gen,
                              "ranks": [{"rank": r, "timed_queries": m, "db_entries": n, "db_digest": d} for r, m, n, d in sizes],
                              "config": {"workload": "dry run: no device work", "targets": int(db.n), "queries_per_step": G}}))
        if world > 1:
            dist.destroy_process_group()
        return

    from foldseek_amd import api
    # host wait policy of the library (fsgpu_ctx.h::syncStream): a waiting feeder thread polls the stream for FSGPU_SPIN_US
    # and then sleeps.  Polling all the way is 1.5 % faster but costs a core per thread; only do it when the cores this job
    # may use (cgroup quota!) comfortably cover every rank's feeder threads.
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    if "FSGPU_SPIN_US" not in os.environ:
        os.environ["FSGPU_SPIN_US"] = "1000000" if usable_cores() >= 2 * local_world * nthreads else "40"
    # host pool for the per-hit backtraces: the cores this rank may use beyond its feeder threads (cgroup quota / local ranks)
    if "FSGPU_HOST_WORKERS" not in os.environ:
        api.set_host_workers(max(0, min(8, usable_cores() // local_world - nthreads)))
    ctx0 = api.Context(local_rank)
    ctx0.adopt_device_db(tensors[0].data_ptr(), tensors[1].data_ptr(), tensors[2].data_ptr(), tensors[3].data_ptr(), db.n, db.data3di.size)
    ctx0._keep = (np.ascontiguousarray(db.data3di), np.ascontiguousarray(db.dataaa), np.ascontiguousarray(db.offsets, np.uint64),
                  np.ascontiguousarray(db.lengths, np.int32))
    del tensors
    par = api.default_params()
    par.alignmentType = args.alignment_type
    # host threads feed the GPU the way the reference's OpenMP threads feed its aligners: each owns a context clone
    # (own HIP stream + scratch) on the shared resident DB; ctypes releases the GIL inside the library calls.
    ctxs = [ctx0] + [ctx0.clone() for _ in range(nthreads - 1)]
    searches = [api.Search(c, par) for c in ctxs]

    def step1(t, i):
        hits = searches[t].prefilter(q3[i])
        res = searches[t].align(qa[i], q3[i], hits["id"])
        return hits, res

    kms, sms, counts = [], [], [0, 0]
    host = {"backtrace_s": 0.0, "rev_pairs": 0.0, "gates_s": 0.0, "profiles_s": 0.0, "sw_wait_s": 0.0}
    lock = threading.Lock()
    ready = threading.Barrier(nthreads + 1)
    go = threading.Barrier(nthreads + 1)
    trace, t_go = [], [0.0]

    def step(t, ids):
        """one step: ONE multi-query scan call for the batch (queries of equal ceil(L / 16) share a launch), then ONE
        multi-query SW launch per pass over all their hit lists"""
        hl = searches[t].prefilter_batch([q3[i] for i in ids])
        scan_ms = ctxs[t].kernel_ms(0)
        launches, nbatched = ctxs[t].gapless_last_batch()
        rs = searches[t].align_batch([qa[i] for i in ids], [q3[i] for i in ids], [h["id"] for h in hl])
        return hl, rs, (scan_ms, launches, nbatched, sum(len(q3[i]) for i in ids))

    # the rank's queries are processed in length order (the order is free; the reference sorts its GPU database by length for
    # the same reason): a batch then spans few register classes, i.e. few scan launches
    timed = sorted(range(n_warm, nq), key=lambda i: len(q3[i]))
    batches = [timed[k:k + G] for k in range(0, len(timed), G)]
    warm = [list(range(k, min(k + G, n_warm))) for k in range(0, n_warm, G)]

    def worker(t):
        # untimed warm-up inside the worker: the first HIP calls of a host thread initialise per-thread state
        for b in warm[t::nthreads]:
            step(t, b)
        if not warm[t::nthreads]:
            step(t, list(range(min(G, nq))))
        # one query of every 16-row length class (one gapless instantiation each; the SW classes are coarser): the first
        # launch of a kernel instantiation (lazy code-object load, attribute set-up, scratch growth) must not land in
        # the timed region
        cls = {}
        for i in range(nq):
            cls.setdefault((len(q3[i]) + 15) // 16, i)
        step(t, sorted(cls.values()))
        ready.wait()
        go.wait()
        for b in batches[t::nthreads]:
            tg = time.perf_counter()
            hl, rs, km = step(t, b)
            st = searches[t].stats()
            with lock:
                if os.environ.get("FS_BENCH_TRACE"):
                    trace.append((t, tg - t_go[0], time.perf_counter() - tg))
                kms.append(km); sms.append(ctxs[t].kernel_ms(1) / len(b))
                counts[0] += sum(len(h) for h in hl); counts[1] += sum(len(r) for r in rs)
                host["profiles_s"] += st[2]; host["sw_wait_s"] += st[3]; host["gates_s"] += st[4]; host["backtrace_s"] += st[5]
                host["rev_pairs"] += st[7]

    ths = [threading.Thread(target=worker, args=(t,)) for t in range(nthreads)]
    for th in ths:
        th.start()
    ready.wait()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    import gc
    gc.collect()
    gc.disable()            # a generation-2 collection of the interpreter (torch is imported: ~50 ms) would stall every feeder thread at once
    t0 = time.perf_counter()
    t_go[0] = t0
    go.wait()
    for th in ths:
        th.join()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = fdist.max_over_ranks(time.perf_counter() - t0, dev)
    gc.enable()
    if os.environ.get("FS_BENCH_TRACE"):
        for rec in sorted(trace, key=lambda r: r[1]):
            print("trace thread %d start %.2f ms dur %.2f ms" % (rec[0], rec[1] * 1e3, rec[2] * 1e3), file=sys.stderr)
    # dominant-kernel duration on an otherwise idle GPU (in the timed region the feeder threads' launches overlap, which
    # stretches each kernel's wall time)
    solo_g, solo_s = [], []
    for i in range(n_warm, min(nq, n_warm + 8)):
        step1(0, i)
        solo_g.append(ctxs[0].kernel_ms(0)); solo_s.append(ctxs[0].kernel_ms(1))
    tot = fdist.gather_objects((counts[0], counts[1], n_mine, host))
    ctx = ctx0

    if rank == 0:
        residues = db.residues
        nq_total = sum(x[2] for x in tot)
        nh, nr = sum(x[0] for x in tot), sum(x[1] for x in tot)
        value = nq_total * residues / dt
        VALU_PEAK = 1024 * 64 * (4.0 / 3.0) / 4.3 * 2.4
        kavg = float(np.mean(solo_g)) * 1e-3
        # scan launches of the timed region: device time of a batch's launches (HIP events on the library's stream) / their number
        n_launch = sum(k[1] for k in kms)
        n_batched = sum(k[2] for k in kms)
        scan_s = sum(k[0] for k in kms) * 1e-3
        kreg = scan_s / max(1, n_launch)                   # average duration of ONE scan launch
        q_per_launch = n_batched / max(1, n_launch)
        lq_timed = [len(q3[i]) for i in timed]
        cells_reg = float(np.mean(lq_timed)) * residues * q_per_launch     # DP cells of one launch
        solo_lq = float(np.mean([len(q3[i]) for i in range(n_warm, min(nq, n_warm + 8))]))
        alg_q = residues + db.n                            # per query: every target residue read once (1 B) + 1 score byte written
        alg_bytes = alg_q * q_per_launch                   # per launch
        mean_lq = float(np.mean(lq_timed))
        cells = solo_lq * residues
        traffic, traffic_src = None, None
        try:   # HBM bytes per launch from a committed PMC pass (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE) of the same kernel on the same DB size
            tj = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            e = tj.get(str(args.targets))
            if e:   # the PMC entries are per query; a launch of the timed region covers q_per_launch of them
                traffic = (e["fetch_correction"] * e["fetch_size_kb"] * 1024 + e["write_size_kb"] * 1024) * q_per_launch
                traffic_src = e["source"]
        except Exception:
            traffic = None
        mine = tot[0][3]
        out = {
            "metric": "residues aligned/sec (prefilter+align)",
            "value": value, "unit": "residues/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f16 (integer-exact, scaled 2^-11) gapless scan + i16 SW", "data": "synthetic",
            "config": {"workload": f"1 step = {G} queries vs {db.n}-structure synthetic 3Di DB (mean len {residues / db.n:.0f}, {args.homologs} planted homologs per query): "
                                   f"gapless prefilter (all targets; queries of one register class share one multi-query scan launch) + top-1000 per query, then one multi-query fwd/rev structure SW launch per pass "
                                   f"(--alignment-type {args.alignment_type}; forward over all pairs, reversed over the pairs that pass the forward gates) "
                                   f"+ host gates + block-aligner backtrace of every accepted hit; {nthreads} host feeder threads per GPU run their steps concurrently",
                       "targets": db.n, "db_residues": residues, "mean_query_len": mean_lq, "query_len_range": [q_lo, q_hi], "max_seqs": 1000, "homologs_per_query": args.homologs,
                       "queries_per_step": G, "queries_total": nq_total, "host_threads_per_gpu": nthreads, "host_backtrace_workers_per_gpu": api.host_workers(),
                       "parallelism": f"query-shard x{world} ({args.scaling}), DB replicated by one RCCL broadcast"},
            "queries_per_s": nq_total / dt, "ms_per_query": 1e3 * dt / (nq_total / world),
            "hits_per_query": nh / nq_total, "alignments_per_query": nr / nq_total,
            # rank 0's host-side accounting of the align leg (sums over its feeder threads, per query)
            "align_leg": {"reverse_pass_fraction": mine["rev_pairs"] / max(1, tot[0][0]),
                          "host_backtrace_ms_per_query": 1e3 * mine["backtrace_s"] / max(1, n_mine),
                          "host_gates_ms_per_query": 1e3 * mine["gates_s"] / max(1, n_mine),
                          "host_profiles_ms_per_query": 1e3 * mine["profiles_s"] / max(1, n_mine),
                          "sw_call_wall_ms_per_query": 1e3 * mine["sw_wait_s"] / max(1, n_mine),
                          "sw_kernels_ms_per_query": float(np.mean(sms)), "sw_kernel_ms_single_query_solo": float(np.mean(solo_s))},
            # per-launch duration of the dominant kernel from HIP events on the library's stream, averaged over the launches
            # of the TIMED region (the host threads overlap their launches there, which stretches each one); the same
            # kernel alone on the device is reported under "solo"
            "roofline": {"bound": "hbm", "achieved": alg_bytes / kreg / 1e9, "peak": 8000.0, "unit": "GB/s",
                         "frac": alg_bytes / kreg / 1e9 / 8000.0, "traffic": traffic, "traffic_source": traffic_src, "algorithmic_bytes": alg_bytes,
                         "kernel": "k_gapless", "kernel_ms": kreg * 1e3, "queries_per_launch": q_per_launch, "launches": n_launch,
                         "note": "the scan is VALU/LDS bound (Lq cell updates per target byte), see valu below and DESIGN.md",
                         # 0.75 packed VALU lane-ops per DP cell (2 x v_pk_add_f16 clamp + 1 x v_pk_maximum3_f16 per 4 cells); these
                         # issue once per 4.3 cycles per SIMD (measured, profiles/r01_valu_lds_issue_rate_ubench.txt):
                         # 1024 SIMDs x 64 lanes x 4/3 cells / 4.3 cyc x 2.4 GHz
                         "valu": {"achieved_gcups": cells_reg / kreg / 1e9, "peak_gcups": VALU_PEAK,
                                  "frac": cells_reg / kreg / 1e9 / VALU_PEAK,
                                  "device_level_gcups": nq_total / world * mean_lq * residues / dt / 1e9,
                                  "device_level_frac": nq_total / world * mean_lq * residues / dt / 1e9 / VALU_PEAK,
                                  "note": "per launch = one multi-query k_gapless launch (scan batches of the feeder threads run one at a time, SW / selection kernels of the other threads co-run); device_level = cells of all timed queries of one rank / wall time"},
                         "solo": {"note": "one single-query launch on an idle device", "kernel_ms": kavg * 1e3, "achieved": alg_q / kavg / 1e9, "frac": alg_q / kavg / 1e9 / 8000.0,
                                  "valu_achieved_gcups": cells / kavg / 1e9, "valu_frac": cells / kavg / 1e9 / VALU_PEAK}},
            "db_broadcast_s": t_bcast, "db_generation_s": t_gen,
        }
        if not args.no_cpu_baseline and world == 1:          # the CPU baselines are an N = 1 item (rank 0 has the host to itself)
            hits, _ = step1(0, n_warm)
            out["cpu_baseline"] = cpu_baseline(db, q3[n_warm], qa[n_warm], hits["id"], args.alignment_type, args.cpu_sample_targets,
                                               [q3[i] for i in range(n_warm + 1, min(nq, n_warm + max(1, args.cpu_sample_queries)))])
    for x in searches[1:]:
        x.close()
    for c in ctxs[1:]:
        c.close()
    searches, ctxs = searches[:1], ctxs[:1]
    kout = None
    if not args.no_kmer:
        kout = kmer_section(args, api, synth, ctx0, searches[0], par, db, rank, world, dev, fdist, [q3[i] for i in timed], [qa[i] for i in timed])
    if rank == 0:
        if kout is not None:
            if not args.no_cpu_baseline and world == 1:
                kout["cpu_baseline"] = kmer_cpu_baseline(args, synth, db)
            out["kmer_prefilter"] = kout
        print(json.dumps(out))
    for x in searches:
        x.close()
    for c in ctxs:
        c.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
