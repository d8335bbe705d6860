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


T_START = time.perf_counter()
# The contract is ONE JSON line on stdout.  Libraries print there too (librccl writes its version banner to the C stdout when the first communicator
# comes up): file descriptor 1 is pointed at stderr for the whole run, and the line goes to a private duplicate of the real stdout (emit_line).
_REAL_STDOUT = None


def _claim_stdout():
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def emit_line(obj):
    sys.stdout.flush()
    data = (json.dumps(obj) + "\n").encode()
    if _REAL_STDOUT is None:
        sys.stdout.write(data.decode()); sys.stdout.flush()
    else:
        while data:                                   # a write to a pipe may be partial
            data = data[os.write(_REAL_STDOUT, data):]


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
    ap.add_argument("--type2-steps", type=int, default=12, help="configs[3] leg of the default run: that many extra steps with --alignment-type 2 (3Di+AA) on the same "
                    "resident DB, reported under `align_type2` (0 = skip; skipped when the main run already is type 2)")
    ap.add_argument("--allvsall-steps", type=int, default=24, help="configs[4] leg of the default run: that many batches of --allvsall-batch DB entries as queries (k-mer prefilter "
                    "+ structurealign on a --allvsall-targets DB), reported under `allvsall` (0 = skip)")
    ap.add_argument("--allvsall-targets", type=int, default=200000)
    ap.add_argument("--allvsall-families", type=int, default=-1, help="configs[4] DB: that many seed structures, each with targets / families - 1 mutated relatives "
                    "(the shape a clustering input has); -1: targets / 10, 0: unrelated structures (only the self match survives -e 0.01)")
    ap.add_argument("--fullrange-steps", type=int, default=8, help="querylen_full_range leg of the default run: that many extra steps with query lengths drawn from the "
                    "DB's own 30..2000 range (their homologs are planted too); 0 disables it")
    ap.add_argument("--single-targets", type=int, default=100000, help="configs[1] leg of the default run (`single_query_100k`): one query at a time against a DB of "
                    "that many structures, end-to-end latency; 0 = skip")
    ap.add_argument("--allvsall-batch", type=int, default=1024, help="queries per device batch of the all-vs-all leg (1024 = the limit of a device batch)")
    ap.add_argument("--emulate-rank-share", type=int, default=0, help="N: on ONE GPU, run what ONE rank of an N-rank node runs: CPU affinity cut to usable_cores / N, "
                    "backtrace pool sized for that, and with --scaling strong only 1/N of the queries (step size adapted like a real rank's)")
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


def cpu_baseline(db, q3, qa, hits_ids, atype, sample_targets, more_queries=(), reuse_prefilter=None):
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
        if reuse_prefilter is not None:          # the gapless prefilter does not depend on the alignment type: one timing serves both legs
            t_pref = reuse_prefilter
        else:
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
    nqk_total = max(32, args.kmer_queries // 32 * 32)
    # strong scaling: the section's queries are a fixed set split over the ranks (at least one batch of 32 each); weak: every rank runs them all
    nqk = max(32, nqk_total // world // 32 * 32) if args.scaling == "strong" else nqk_total
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
        btc = ksearch[t].backtrace_counts()
        if timed:
            with lock:
                stat["bt_device"] = stat.get("bt_device", 0) + btc[0]; stat["bt_all"] = stat.get("bt_all", 0) + btc[1]
                stat["dev"].append(ms[0]); stat["lists"].append(ms[10]); stat["counts"].append(cnt); stat.setdefault("sw", []).append(swms)
                stat["hits"] += sum(len(r) for r in res); stat["aln"] += na; stat["t_pref"] += tp; stat["t_aln"] += ta
                stat["bad"] = stat.get("bad", 0) + int((status < 0).sum())
        return res

    for t in range(KT):
        run(t, batches[0], False)                                 # warm every host thread's context
    ready, go = threading.Barrier(KT + 1), threading.Barrier(KT + 1)

    failed = []

    def worker(t):
        ready.wait(); go.wait()
        try:
            for b in range(t, len(batches), KT):
                run(t, batches[b], True)
        except BaseException as e:          # reported after the join: a leg with a dead feeder thread must not print partial numbers
            failed.append(e)

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
    if failed:
        raise RuntimeError(f"k-mer leg: a feeder thread failed: {failed[0]!r}") from failed[0]
    # solo batch on an idle GPU for the roofline of the dominant kernel: the fastest of three (the first one can still overlap the last SW
    # launches of the timed region)
    solo_ms = None
    for _ in range(3):
        torch.cuda.synchronize()
        run(0, batches[-1], False)
        ms_i = kctx[0].kmer_stage_ms()
        if solo_ms is None or ms_i[0] < solo_ms[0]:
            solo_ms, solo_cnt = ms_i, kctx[0].kmer_counts()
    for x in ksearch[1:]:
        x.close()
    for c in kctx[1:]:
        c.close()
    if rank != 0:
        return None
    # Roofline of the device part of ONE prefilter batch (32 queries, every k_kmer_* kernel from the similar-k-mer lists to the selected
    # results, HIP events on the library's stream).  Algorithmic bytes: 8 per similar k-mer (the two u32 offsets of its probe)
    # + 8 per index hit (the entry gathered) + the target residues under every scored diagonal.  "traffic": FETCH_SIZE + WRITE_SIZE of all
    # k_kmer_* kernels of such a batch from a committed --pmc pass, per index hit, times this batch's hits.
    qlen = float(np.mean([len(q) for q in q3]))

    def alg_of(cnt):
        return 8.0 * float(cnt[0]) + 8.0 * float(cnt[1]) + float(cnt[2]) * qlen

    reg_ms = float(np.mean(stat["dev"]))
    reg_cnt = np.mean(np.asarray(stat["counts"], dtype=np.float64), axis=0)
    reg_alg, alg = alg_of(reg_cnt), alg_of(solo_cnt)
    e, traffic_src = pmc_traffic_entry(os.path.join(ROOT, "profiles", "pmc_traffic_kmer.json"), args.targets)
    per_hit = None if e is None else e.get("k_kmer_all_bytes_per_index_hit")
    lists_per_probe = None if e is None else e.get("k_kmer_lists_bytes_per_probe")
    probes, lists_s = float(solo_cnt[0]), solo_ms[10] * 1e-3
    out = {"workload": f"{nqk} queries in batches of 32 vs the same {db.n}-structure DB: k-mer prefilter (-s 9.5, k=6 spaced, "
                       f"--max-seqs 1000, double-diagonal + ungapped scoring) + fwd/rev structure SW on its hits (one multi-query SW launch per register class)",
           "value": world * nqk * db.residues / dt, "unit": "residues/s", "queries_per_s": world * nqk / dt,
           "ms_per_query": 1e3 * dt / nqk, "prefilter_ms_per_query_host_wall": 1e3 * stat["t_pref"] / nqk,
           "align_ms_per_query_host_wall": 1e3 * stat["t_aln"] / nqk, "sw_kernels_ms_per_batch32": float(np.mean(stat["sw"])), "prefilter_device_ms_per_query": float(np.sum(stat["dev"])) / nqk,
           "prefilter_device_ms_per_query_solo": solo_ms[0] / 32,
           "host_threads": KT, "index_build_s": t_index, "index_entries": int(ctx0.kmer_index_entries), "kmer_threshold": thr,
           "similar_kmers_per_query": float(reg_cnt[0]) / 32, "index_hits_per_query": float(reg_cnt[1]) / 32,
           "candidates_per_query": float(reg_cnt[2]) / 32,
           "hits_per_query": stat["hits"] / nqk, "alignments_per_query": stat["aln"] / nqk, "unsupported_queries": stat.get("bad", 0),
           "backtraces_on_device": [int(stat.get("bt_device", 0)), int(stat.get("bt_all", 0))],
           "stage_ms_per_batch32_solo": {k: solo_ms[i] for i, k in enumerate(["device_total", "count", "lists", "emit", "partition", "dup", "score", "replay", "select", "host_tail", "k_kmer_lists"])},
           "segments_solo": {k: int(v) for k, v in zip(["unused0", "runs", "unused2", "tiles", "runs_all", "coarse_keys", "ids_of_widest_key"], kctx[0].kmer_segments()) if not k.startswith("unused")},
           # kernel_ms: mean over the batches of the timed region (the host threads overlap their batches and the SW launches);
           # "solo" = the same batch alone on the device
           "roofline": {"bound": "hbm", "kernel": "k_kmer_* (the device part of one prefilter batch of 32 queries, all kernels)", "kernel_ms": reg_ms,
                        "achieved": reg_alg / (reg_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                        "frac": reg_alg / (reg_ms * 1e-3) / 1e9 / 8000.0, "traffic": None if per_hit is None else per_hit * float(reg_cnt[1]), "traffic_source": traffic_src,
                        "algorithmic_bytes": reg_alg, "index_hits_per_launch": float(reg_cnt[1]), "index_hits_per_s": float(reg_cnt[1]) / (reg_ms * 1e-3),
                        "solo": {"kernel_ms": solo_ms[0], "achieved": alg / (solo_ms[0] * 1e-3) / 1e9, "frac": alg / (solo_ms[0] * 1e-3) / 1e9 / 8000.0,
                                 "algorithmic_bytes": alg, "index_hits_per_launch": float(solo_cnt[1]), "index_hits_per_s": float(solo_cnt[1]) / (solo_ms[0] * 1e-3),
                                 "traffic": None if per_hit is None else per_hit * float(solo_cnt[1])},
                        "k_kmer_lists_solo": {"kernel_ms": solo_ms[10], "algorithmic_bytes": probes * 8.0, "achieved": probes * 8.0 / max(lists_s, 1e-12) / 1e9,
                                              "frac": probes * 8.0 / max(lists_s, 1e-12) / 1e9 / 8000.0, "probes_per_s": probes / max(lists_s, 1e-12),
                                              "traffic": None if lists_per_probe is None else lists_per_probe * probes},
                        "note": "bytes = 8 per similar k-mer + 8 per index hit + diagonal residues; the hit stream itself (8-byte records: written by the emit, "
                                "moved twice by the two-level partition, read by the duplicate-diagonal kernels) is pipeline traffic, not algorithmic, "
                                "so frac is bounded by about 8 / (8 + 5 * 8) even at HBM speed, see DESIGN.md 4.5"}}
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


def sw_traffic_per_pair(targets, has_aa):
    """HBM bytes per target pair of the batch SW from the committed PMC pass (profiles/pmc_traffic_sw.json), or (None, reason)"""
    at = "2" if has_aa else "0"
    path = os.path.join(ROOT, "profiles", "pmc_traffic_sw.json")
    try:
        have = json.load(open(path))
    except Exception:
        have = {}
    key = f"{targets}:{at}" if f"{targets}:{at}" in have else targets          # round 6: one entry per (DB size, alignment type); before: per DB size
    e, src = pmc_traffic_entry(path, key)
    if e is None:
        return None, src
    if str(e.get("alignment_type")) != at:
        return None, f"the PMC pass ran --alignment-type {e.get('alignment_type')}"
    return float(e["bytes_per_pair"]), src


def sw_roofline(passes, has_aa, solo=None, targets=None):
    """issue-rate roofline of k_sw2: a wave-instruction of the DP row loop updates 64 lanes x 2 int16 halves = 128 cells and the
    row costs 14 packed VALU instructions (15 with the AA table); packed 16-bit ops issue once per 4.3 cycles per SIMD (measured,
    profiles/r01_valu_lds_issue_rate_ubench.txt) -> 1024 SIMDs x 128 / 14 / 4.3 cyc x 2.4 GHz.  `passes` = fsgpu_sw_last_passes() arrays
    of the timed batches (forward + reversed-query pass each); achieved = DP cells of those passes / their device time (HIP events)."""
    per = 15.0 if has_aa else 14.0          # k_sw3: the AA table is added on packed row pairs, one v_pk_add_i16 per row (k_sw2: two)
    peak = 1024 * 128 / per / 4.3 * 2.4          # Gcell/s
    cells = sum(float(p[d][1]) for p in passes for d in (0, 1) if p[d][0] >= 0)
    ms = sum(float(p[d][0]) for p in passes for d in (0, 1) if p[d][0] >= 0)
    instr = sum(float(p[d][3]) for p in passes for d in (0, 1) if p[d][0] >= 0)
    out = {"bound": "valu-issue", "kernel": "k_sw3", "unit": "Gcell/s", "peak": peak, "achieved": cells / max(ms, 1e-9) / 1e6,
           "frac": cells / max(ms, 1e-9) / 1e6 / peak, "cells_per_pass_pair": cells / max(1, len(passes)), "kernel_ms_per_pass_pair": ms / max(1, len(passes)),
           "traffic": None,
           "pairs_per_pass_pair": sum(float(p[d][2]) for p in passes for d in (0, 1) if p[d][0] >= 0) / max(1, len(passes)),
           # the same passes priced in what the waves really issue: a wave (four targets at 32 lanes per pair, two at 64) runs (its longest
           # target) + lanes - 1 steps of 14 R + 16 VALU instructions (+ R + 4 with the AA table; counted in the kernel's ISA), 4.3 cycles each.
           # frac / issued_valu_frac = the share of the issued stream that is the 14 (15) DP instructions of real cells: the rest is the step
           # overhead, lane padding, wavefront fill / drain and the shorter targets of a wave
           "issued_valu_frac": instr / max(ms, 1e-9) / 1e-3 / (1024 * 2.4e9 / 4.3),
           "note": "co-running with the other feeder threads' scans; DP cells = query rows x target columns of every pair of both passes"}
    if targets is not None:
        per_pair, src = sw_traffic_per_pair(targets, has_aa)
        out["traffic_source"] = src
        if per_pair is not None:
            out["traffic"] = per_pair * out["pairs_per_pass_pair"]        # HBM bytes of one forward + reversed pass pair (PMC bytes per target pair x its pairs)
    if solo is not None:
        c = sum(float(solo[d][1]) for d in (0, 1) if solo[d][0] >= 0)
        m = sum(float(solo[d][0]) for d in (0, 1) if solo[d][0] >= 0)
        si = sum(float(solo[d][3]) for d in (0, 1) if solo[d][0] >= 0)
        out["solo"] = {"note": "one batch alone on the device", "kernel_ms": m, "achieved": c / max(m, 1e-9) / 1e6, "frac": c / max(m, 1e-9) / 1e6 / peak,
                       "issued_valu_frac": si / max(m, 1e-9) / 1e-3 / (1024 * 2.4e9 / 4.3)}
    return out


def allvsall_cpu_baseline(db, thr, sample_queries=64):
    """reference k-mer prefilter (-s 4.5 --max-seqs 200) + reference structurealign (3Di+AA, -e 0.01 -c 0.8) on a sample of the DB's own entries"""
    import kmer_lib as K
    import oracle_lib
    R = K.load_ref()
    ref = oracle_lib.load_ref()
    if R is None or ref is None:
        return None
    threads = usable_cores()
    ids = np.linspace(0, db.n - 1, sample_queries).astype(np.int64)
    q3 = [db.seq(int(i), "3di") for i in ids]
    qa = [db.seq(int(i), "aa") for i in ids]
    t0 = time.perf_counter()
    r = K.RefKpf(R, [db.seq(i, "3di", unmask=False) for i in range(db.n)], threads=threads, kmerThr=thr, maxResListLen=200)
    t_build = time.perf_counter() - t0
    res, _, secs = r.run(q3, ids, threads=threads)
    res, _, secs = r.run(q3, ids, threads=threads)
    r.close()
    t3 = np.where(db.data3di >= 32, db.data3di - 32, db.data3di).astype(np.uint8)
    t_aln = 0.0
    for q, a, hits in zip(q3, qa, res):
        lt = db.lengths[hits["id"]].astype(np.float32)
        lq = np.float32(len(q))
        h = np.ascontiguousarray(hits["id"][(lq / lt >= 0.8) & (lt / lq >= 0.8)].astype(np.int64))
        if not len(h):
            continue
        aln = np.zeros(len(h), oracle_lib.REFALN_DT)
        t_aln += ref.ref_structure_align(a, q, len(q), 2, 0, 0.5, 10, 1, db.dataaa, t3, np.ascontiguousarray(db.offsets[:-1][h]), np.ascontiguousarray(db.lengths[h]),
                                         len(h), db.residues, 0.01, 1, threads, None, None, aln.ctypes.data, None, 0)
    tot = secs + t_aln
    return {"value": sample_queries * db.residues / tot, "unit": "residues/s", "queries_per_s": sample_queries / tot, "cores": threads, "kind": "reference",
            "sample": f"{sample_queries} DB entries spread over the length range as queries: QueryMatcher::matchQuery of the reference (k-mer threshold {thr}, --max-seqs 200, "
                      f"{threads} OpenMP threads, second of two runs, index build {t_build:.1f}s not included) + the reference's alignStructure (3Di+AA, no composition bias correction) "
                      f"on the coverage-filtered hits, one query after the other with {threads} threads over its hits",
            "prefilter_s": secs, "align_s": t_aln, "index_build_s": t_build}


def allvsall_db(synth, targets, families):
    """configs[4] input: `families` seed structures + targets / families - 1 mutated relatives of each (substitution rate 20..60 %, indels:
    synth._homologs), the shape a clustering input has; families == 0: unrelated structures."""
    if families <= 0:
        return synth.make_db_fast(targets, None, seed=20260923, homologs_per_query=0)
    sd = synth.make_db_fast(families, None, seed=7, homologs_per_query=0, mask_frac=0, x_frac=0)
    seeds = ([sd.data3di[sd.offsets[i]:sd.offsets[i] + sd.lengths[i]].copy() for i in range(families)],
             [sd.dataaa[sd.offsets[i]:sd.offsets[i] + sd.lengths[i]].copy() for i in range(families)])
    return synth.make_db_fast(targets, seeds, seed=20260923, homologs_per_query=max(targets // families - 1, 1))


def allvsall_run(args, api, synth, fdist, dev, rank, world, local_rank, targets, steps, warmup, with_cpu=False):
    """configs[4]: all-vs-all of a `targets`-structure DB, the shape of easy-cluster's cascaded steps (F/data/structurecluster.sh via
    `easy-cluster -v 3`): prefilter -s 4.5 --max-seqs 200 --min-ungapped-score 30 -c 0.8 --add-self-matches 1, then structurealign
    -e 0.01 -c 0.8 --comp-bias-corr 0.  Prefilter dominated: one pass of the k-mer index per batch of 32 queries.  Every DB entry is a
    query; the ids shard over the ranks (`steps` batches each, 0 = all of them), the DB is replicated by one broadcast.  Returns the
    result object on rank 0 (None elsewhere)."""
    import threading
    import torch
    import torch.distributed as dist
    t_gen = time.perf_counter()
    fam = args.allvsall_families if args.allvsall_families >= 0 else targets // 10
    db = allvsall_db(synth, targets, fam) if rank == 0 else None
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
    AB = max(32, args.allvsall_batch // 32 * 32)        # queries per device batch: the low-sensitivity prefilter yields ~10^4 index hits per query, far too few to fill the device in 32s
    strong = args.scaling == "strong" and steps > 0 and world > 1
    if strong:
        # the SAME `steps` timed batches whatever the number of ranks: picked over the whole DB, dealt out to the ranks round robin
        ids = np.arange(db.n)
        nb_all = (db.n + AB - 1) // AB
        nb = min(nb_all, steps + warmup * world)
        pick = np.linspace(0, nb_all - 1, nb).astype(np.int64) if nb < nb_all else np.arange(nb_all)
        pick = pick[rank::world]
    else:
        lo, hi = fdist.shard_range(db.n, rank, world)
        ids = np.arange(lo, hi)
        nb_all = (len(ids) + AB - 1) // AB
        nb = nb_all if steps <= 0 else min(nb_all, steps + warmup)
        # a spread sample of the shard when only some batches are run: the DB is length sorted
        pick = np.linspace(0, nb_all - 1, nb).astype(np.int64) if nb < nb_all else np.arange(nb_all)
    batches = [ids[b * AB:(b + 1) * AB] for b in pick]
    if steps > 0:            # warm-up batches spread over the length range as well, so every register class has been launched once
        wsel = set(np.linspace(0, len(batches) - 1, min(warmup, len(batches))).astype(np.int64).tolist()) if warmup > 0 else set()
        warm = [b for i, b in enumerate(batches) if i in wsel]
        timed = [b for i, b in enumerate(batches) if i not in wsel]
    else:
        warm, timed = batches[:1], batches
    stat = {"hits": 0, "aln": 0, "q": 0, "res": 0, "dev": 0.0, "bad": 0, "stage": np.zeros(11), "cnt": np.zeros(4), "nb": 0, "swp": [], "diag_bytes": 0.0}
    lock = threading.Lock()

    # the queries are DB entries: unmasked code strings once, then a batch is two arrays of host addresses -- no per-query work in the interpreter;
    # profiles, device prefilter, coverage pre-filter and alignment of a batch are ONE library call (fshost_search_kmer_batch, the per-batch body
    # of `fsgpu-modules search`)
    u3 = np.ascontiguousarray(np.where(db.data3di >= 32, db.data3di - 32, db.data3di).astype(np.uint8))
    ua = np.ascontiguousarray(np.where(db.dataaa >= 32, db.dataaa - 32, db.dataaa).astype(np.uint8))
    base3, baseA = np.uint64(u3.ctypes.data), np.uint64(ua.ctypes.data)
    offs64 = np.ascontiguousarray(db.offsets[:-1], np.uint64)
    lens32 = np.ascontiguousarray(db.lengths, np.int32)

    def run(t, b, count):
        tw0 = time.perf_counter()
        bb = np.asarray(b, np.int64)
        ls = lens32[bb]
        tw1 = time.perf_counter()
        r = searches[t].kmer_batch(m8, m2, thr, 1, baseA + offs64[bb], base3 + offs64[bb], ls, pref_identity=bb, aln_identity=bb, max_res=200)
        tw2 = time.perf_counter()
        ms, cnt = ctxs[t].kmer_stage_ms(), ctxs[t].kmer_counts()
        swp = ctxs[t].sw_last_passes()
        btc = searches[t].backtrace_counts()
        if count:
            sec = r["seconds"]
            with lock:
                stat["bt_device"] = stat.get("bt_device", 0) + btc[0]; stat["bt_all"] = stat.get("bt_all", 0) + btc[1]
                stat["hits"] += int(r["nkept"].sum()); stat["aln"] += int(r["nres"].sum()); stat["q"] += len(bb)
                stat["res"] += int(ls.sum()); stat["dev"] += ms[0]; stat["bad"] += int((r["status"] < 0).sum())
                stat["stage"] += np.array(ms[:11]); stat["cnt"] += np.array(cnt, float); stat["nb"] += 1; stat["swp"].append(swp)
                stat["diag_bytes"] += float(cnt[2]) * float(ls.mean())
                w = [(tw1 - tw0) + sec[0], sec[1], sec[2], sec[3], (tw2 - tw1) - float(sec.sum())]
                stat["wall"] = [a + x for a, x in zip(stat.get("wall", [0.0] * 5), w)]

    for t in range(KT):
        for b in warm[t::KT] or warm[:1]:
            run(t, b, False)
    ready, go = threading.Barrier(KT + 1), threading.Barrier(KT + 1)

    failed = []

    def worker(t):
        ready.wait(); go.wait()
        try:
            for b in timed[t::KT]:
                run(t, b, True)
        except BaseException as e:          # reported after the join: a leg with a dead feeder thread must not print partial numbers
            failed.append(e)

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
    if failed:
        raise RuntimeError(f"all-vs-all leg: a feeder thread failed: {failed[0]!r}") from failed[0]
    # one batch alone on the device (the middle one of this rank's timed batches): the solo figures of both rooflines
    solo = None
    if timed:
        keep_stat = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in stat.items()}
        best, best_sw = None, None
        sw_ms = lambda p: sum(float(p[d][0]) for d in (0, 1) if p[d][0] >= 0)
        for _ in range(5):
            torch.cuda.synchronize()
            run(0, timed[len(timed) // 2], False)
            ms_i, cnt_i, swp_i = ctxs[0].kmer_stage_ms(), ctxs[0].kmer_counts(), ctxs[0].sw_last_passes()
            if best is None or ms_i[0] < best[0][0]:
                best = (ms_i, cnt_i, swp_i)
            if best_sw is None or sw_ms(swp_i) < sw_ms(best_sw):
                best_sw = swp_i                                # the SW passes' own fastest repetition, not the one of the fastest k-mer batch
        solo = (best[0], best[1], best_sw)
        stat.update(keep_stat)
    stat["stage"] = stat["stage"].tolist(); stat["cnt"] = stat["cnt"].tolist()
    mine_swp = stat.pop("swp")
    tot = fdist.gather_objects(stat)
    out = None
    if rank == 0:
        nq = sum(x["q"] for x in tot)
        s0 = tot[0]
        nb0 = max(1, s0["nb"])
        names = ["device_total", "count", "lists", "emit", "partition", "dup", "score", "replay", "select", "host_tail", "k_kmer_lists"]
        # HBM roofline of the k-mer prefilter batch (rank 0's batches, HIP events over the whole device part of a batch): algorithmic bytes =
        # 8 per similar k-mer (two u32 offsets of the probe) + 8 per index hit (the entry gathered) + the target residues under every scored diagonal
        alg = 8.0 * s0["cnt"][0] + 8.0 * s0["cnt"][1] + s0["diag_bytes"]
        dev_s = s0["stage"][0] * 1e-3
        e, traffic_src = pmc_traffic_entry(os.path.join(ROOT, "profiles", "pmc_traffic_allvsall.json"), targets)
        per_hit = None if e is None else e.get("k_kmer_all_bytes_per_index_hit")
        solo_obj = None
        if solo is not None:
            mean_l = s0["res"] / max(1, s0["q"])
            salg = 8.0 * float(solo[1][0]) + 8.0 * float(solo[1][1]) + float(solo[1][2]) * mean_l
            solo_obj = {"note": "one batch alone on the device", "kernel_ms": float(solo[0][0]), "algorithmic_bytes": salg,
                        "achieved": salg / max(float(solo[0][0]) * 1e-3, 1e-12) / 1e9, "frac": salg / max(float(solo[0][0]) * 1e-3, 1e-12) / 1e9 / 8000.0,
                        "index_hits_per_launch": float(solo[1][1]), "traffic": None if per_hit is None else per_hit * float(solo[1][1]),
                        "stage_ms": {k: float(solo[0][i]) for i, k in enumerate(names)}}
        out = {"metric": "residues aligned/sec (prefilter+align)", "value": nq * db.residues / dt, "unit": "residues/s", "n_gpus": world,
               "steps": len(timed), "warmup": len(warm), "ms_per_step": 1e3 * dt / max(1, len(timed)), "higher_is_better": True,
               "scaling": "strong" if (steps <= 0 or strong) else "weak", "vs_baseline": None, "dtype": "u8 k-mer index probes / diagonal scores + i16 SW", "data": "synthetic",
               "config": {"workload": f"all-vs-all (configs[4]): {nq} of the {db.n} DB entries as queries in batches of {AB} (1 step = 1 batch), k-mer prefilter -s 4.5 --max-seqs 200 "
                                      f"-c 0.8 + structurealign -e 0.01 -c 0.8 (3Di+AA) on its hits; queries shard over {world} rank(s), DB replicated by one broadcast",
                          "targets": db.n, "db_residues": db.residues, "host_threads_per_gpu": KT, "kmer_threshold": thr, "families": fam,
                          "family_size": (targets // fam) if fam > 0 else 1},
               "queries_per_s": nq / dt, "ms_per_query": 1e3 * dt / max(1, nq / world), "hits_per_query": sum(x["hits"] for x in tot) / max(1, nq),
               "alignments_per_query": sum(x["aln"] for x in tot) / max(1, nq), "prefilter_device_ms_per_query": s0["dev"] / max(1, s0["q"]),
               "unsupported_queries": sum(x["bad"] for x in tot), "index_build_s": t_index, "db_generation_s": t_gen,
               "backtraces_on_device": [int(sum(x.get("bt_device", 0) for x in tot)), int(sum(x.get("bt_all", 0) for x in tot))],
               "projected_full_all_vs_all_s": db.n / max(1e-9, nq / dt),
               "queries_per_batch": AB, "stage_ms_per_batch": {k: s0["stage"][i] / nb0 for i, k in enumerate(names)},
               # wall time of one feeder thread per batch (KT threads run their batches concurrently), all inside ONE fshost_search_kmer_batch call since round 5:
               # query profiles (fshost_kmer_query_prepare in C++), the fsgpu_kmer_search call, the coverage pre-filter, the align call; what the ctypes wrapper adds
               "host_wall_ms_per_batch": {k: 1e3 * v / nb0 for k, v in zip(["prepare", "kmer_search_call", "coverage_filter", "align_call", "ctypes_marshalling"], s0.get("wall", [0.0] * 5))},
               "similar_kmers_per_query": s0["cnt"][0] / max(1, s0["q"]), "index_hits_per_query": s0["cnt"][1] / max(1, s0["q"]),
               "candidates_per_query": s0["cnt"][2] / max(1, s0["q"]),
               "roofline": {"bound": "hbm", "kernel": f"k_kmer_* (the device part of one prefilter batch of {AB} queries, all kernels)", "unit": "GB/s", "peak": 8000.0,
                            "achieved": alg / max(dev_s, 1e-12) / 1e9, "frac": alg / max(dev_s, 1e-12) / 1e9 / 8000.0,
                            "traffic": None if per_hit is None else per_hit * s0["cnt"][1] / nb0, "traffic_source": traffic_src,
                            "algorithmic_bytes": alg / nb0, "kernel_ms": s0["stage"][0] / nb0, "solo": solo_obj,
                            "note": "co-running with the other feeder threads' batches and SW launches; bytes = 8 per similar k-mer + 8 per index hit + diagonal residues"},
               "align_roofline": sw_roofline(mine_swp, True, None if solo is None else solo[2], targets)}
        if with_cpu:
            out["cpu_baseline"] = allvsall_cpu_baseline(db, thr)
    for x in searches:
        x.close()
    for c in ctxs[1:]:
        c.close()
    ctx0.close()
    if out is not None and world == 1 and with_cpu:
        # the same configuration through the product's own host code: the C++ module `fsgpu-modules search` on the same DB written to disk,
        # ALL its entries as queries, wall time of the whole process (start, DB read, index build, every query, result write).  The leg above
        # feeds the C ABI from Python threads (sequence extraction, ctypes marshalling and the coverage pre-filter run under the interpreter
        # lock: `host_wall_ms_per_batch`), this is what a user of the module gets.
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import allvsall_modules
            out["native_module_end_to_end"] = allvsall_modules.run(threads=usable_cores(), db=db, fam=fam)
        except Exception as e:                                             # noqa: BLE001 -- the leg's own numbers stand without it
            out["native_module_end_to_end"] = {"error": str(e)[-500:]}
    return out


def single_query_leg(args, api, synth, local_rank, with_cpu):
    """configs[1]: ONE query against a 100k-structure DB on one GPU -- the single-query call shape of the reference's ungappedprefilter /
    structurealign loops (M/src/prefiltering/ungappedprefilter.cpp:41-326,346-480): query profile upload, gapless scan of every target,
    selection of the top 1000, forward and reversed structure SW, host gates and block-aligner backtraces, nothing batched over queries.
    Reported as LATENCY (ms per query, end to end and per part) next to the reference's own CPU code for the same query on the same DB."""
    import torch
    NQ = 8
    q3, qa = synth.make_queries(NQ, seed=4100, lo=250, hi=450)
    db = synth.make_db_fast(args.single_targets, (q3, qa), seed=20260924, homologs_per_query=args.homologs)
    ctx = api.Context(local_rank)
    ctx.load_db(db)
    par = api.default_params()
    par.alignmentType = args.alignment_type
    search = api.Search(ctx, par)
    rows = []
    for rep in range(4):                       # the first round warms the kernels' register classes and the scratch buffers
        for i in range(NQ):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            hits = search.prefilter(q3[i])
            t1 = time.perf_counter()
            res = search.align(qa[i], q3[i], hits["id"])
            t2 = time.perf_counter()
            if rep:
                rows.append((i, 1e3 * (t2 - t0), 1e3 * (t1 - t0), 1e3 * (t2 - t1), ctx.kernel_ms(0), ctx.kernel_ms(1), len(hits), len(res)))
    a = np.array([r[1:] for r in rows], dtype=np.float64)
    med = np.median(a, axis=0)
    lq = float(np.mean([len(q) for q in q3]))
    VALU_PEAK43 = 1024 * 64 * (4.0 / 3.0) / 4.3 * 2.4
    VALU_PEAK40 = 1024 * 64 * (4.0 / 3.0) / 4.0 * 2.4
    gcups = lq * db.residues / (med[3] * 1e-3) / 1e9
    stripes = (db.n + 7) // 8
    out = {"workload": f"configs[1]: 1 query (lengths 250..450, {args.homologs} planted homologs each; {NQ} queries x 3 repeats, one at a time on an idle device) vs a "
                       f"{db.n}-structure synthetic 3Di DB (mean len {db.residues / db.n:.0f}): gapless prefilter of every target + top-1000 + fwd/rev structure SW "
                       f"(--alignment-type {args.alignment_type}) + gates + backtraces, the single-query C ABI calls (fshost_search_prefilter / fshost_search_align)",
           "targets": int(db.n), "db_residues": int(db.residues), "mean_query_len": lq,
           "end_to_end_ms": float(med[0]), "end_to_end_ms_min": float(a[:, 0].min()), "end_to_end_ms_max": float(a[:, 0].max()),
           "prefilter_call_ms": float(med[1]), "align_call_ms": float(med[2]), "scan_kernel_ms": float(med[3]), "sw_kernels_ms": float(med[4]),
           "hits_per_query": float(a[:, 5].mean()), "alignments_per_query": float(a[:, 6].mean()),
           "value": db.residues / (med[0] * 1e-3), "unit": "residues/s", "queries_per_s": 1e3 / med[0],
           "scan": {"stripes_of_8_targets": int(stripes), "waves_if_one_stripe_each": int(stripes), "wave_slots_of_the_device": 256 * 12,
                    "note": "one query at 100k targets is 12.5k stripes = 12.5k wave-sized work items for 3072 resident waves (256 CUs x 3 workgroups x 4 waves): "
                            "four rounds of waves, the kernel's fill / drain and the launch latency are what a single query pays over the multi-query scan",
                    "achieved_gcups": gcups, "frac_of_valu_bound_4_3cyc": gcups / VALU_PEAK43, "frac_of_valu_bound_ideal_4cyc": gcups / VALU_PEAK40,
                    "hbm_frac": (db.residues + db.n) / (med[3] * 1e-3) / 1e9 / 8000.0}}
    if with_cpu:
        hits = search.prefilter(q3[0])
        cb = cpu_baseline(db, q3[0], qa[0], hits["id"], args.alignment_type, int(db.n), [q3[i] for i in range(1, NQ)])
        t_cpu = cb["prefilter_s_sample"] + cb["align_s"]
        out["cpu_baseline"] = {"value": db.residues / t_cpu, "unit": "residues/s", "end_to_end_ms": 1e3 * t_cpu, "prefilter_ms": 1e3 * cb["prefilter_s_sample"],
                               "align_ms": 1e3 * cb["align_s"], "cores": cb["cores"], "kind": cb["kind"],
                               "sample": f"the reference's runFilterOnCpu scan over ALL {db.n} targets (mean of {NQ} queries) + alignStructure over one query's hit list, {cb['cores']} threads"}
    search.close()
    ctx.close()
    return out


def search_region(api, ctxs, searches, q3, qa, warm_batches, batches, world, dev, fdist, class_warm, nq_all):
    """The timed region of one search leg: every feeder thread (own context clone = own HIP stream, shared resident DB) runs its share of
    `batches`; a step = ONE multi-query scan call for the batch (queries of equal ceil(L / 16) share a launch), then ONE multi-query SW
    launch per pass over all their hit lists.  Bracketed by barrier + device synchronisation on both sides, max over ranks."""
    import gc
    import threading
    import torch
    import torch.distributed as dist
    nthreads = len(ctxs)
    rec = {"kms": [], "sms": [], "counts": [0, 0], "swp": [], "trace": [],
           "host": {"backtrace_s": 0.0, "rev_pairs": 0.0, "gates_s": 0.0, "profiles_s": 0.0, "sw_wait_s": 0.0, "bt_device": 0, "bt_all": 0}}
    lock = threading.Lock()
    ready = threading.Barrier(nthreads + 1)
    go = threading.Barrier(nthreads + 1)
    t_go = [0.0]

    def step(t, ids):
        hl = searches[t].prefilter_batch([q3[i] for i in ids])
        scan_ms = ctxs[t].kernel_ms(0)
        launches, nbatched = ctxs[t].gapless_last_batch()
        rs = searches[t].align_batch([qa[i] for i in ids], [q3[i] for i in ids], [h["id"] for h in hl])
        return hl, rs, (scan_ms, launches, nbatched, sum(len(q3[i]) for i in ids))

    failed = []

    def worker(t):
        try:
            worker_body(t)
        except BaseException as e:          # a feeder thread that dies must not leave the others (and the main thread) waiting at a barrier for good
            failed.append(e)
            ready.abort(); go.abort()

    def worker_body(t):
        # untimed warm-up inside the worker: the first HIP calls of a host thread initialise per-thread state
        for b in warm_batches[t::nthreads]:
            step(t, b)
        if not warm_batches[t::nthreads]:
            step(t, (warm_batches[0] if warm_batches else batches[0]))
        if class_warm:
            # one query of every 16-row length class (one gapless instantiation each; the SW classes are coarser): the first
            # launch of a kernel instantiation (lazy code-object load, attribute set-up, scratch growth) must not land in
            # the timed region
            cls = {}
            for i in range(nq_all):
                cls.setdefault((len(q3[i]) + 15) // 16, i)
            step(t, sorted(cls.values()))
        ready.wait()
        go.wait()
        for b in batches[t::nthreads]:
            tg = time.perf_counter()
            hl, rs, km = step(t, b)
            st = searches[t].stats()
            swp = ctxs[t].sw_last_passes()
            btc = searches[t].backtrace_counts()
            with lock:
                if os.environ.get("FS_BENCH_TRACE"):
                    rec["trace"].append((t, tg - t_go[0], time.perf_counter() - tg))
                rec["kms"].append(km); rec["sms"].append(ctxs[t].kernel_ms(1) / len(b)); rec["swp"].append(swp)
                rec["counts"][0] += sum(len(h) for h in hl); rec["counts"][1] += sum(len(r) for r in rs)
                h = rec["host"]
                h["profiles_s"] += st[2]; h["sw_wait_s"] += st[3]; h["gates_s"] += st[4]; h["backtrace_s"] += st[5]; h["rev_pairs"] += st[7]
                h["bt_device"] += btc[0]; h["bt_all"] += btc[1]

    ths = [threading.Thread(target=worker, args=(t,)) for t in range(nthreads)]
    for th in ths:
        th.start()

    def wait_at(barrier):
        try:
            barrier.wait()
        except threading.BrokenBarrierError:
            for th in ths:
                th.join()
            raise RuntimeError(f"bench.py: a feeder thread failed: {failed[0]!r}" if failed else "bench.py: feeder barrier broken") from (failed[0] if failed else None)

    wait_at(ready)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    gc.collect()
    gc.disable()            # a generation-2 collection of the interpreter (torch is imported: ~50 ms) would stall every feeder thread at once
    t0 = time.perf_counter()
    t_go[0] = t0
    wait_at(go)
    for th in ths:
        th.join()
    if failed:
        gc.enable()
        raise RuntimeError(f"bench.py: a feeder thread failed: {failed[0]!r}") from failed[0]
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = fdist.max_over_ranks(time.perf_counter() - t0, dev)
    gc.enable()
    if os.environ.get("FS_BENCH_TRACE"):
        for r in sorted(rec["trace"], key=lambda r: r[1]):
            print("trace thread %d start %.2f ms dur %.2f ms" % (r[0], r[1] * 1e3, r[2] * 1e3), file=sys.stderr)
    return dt, rec, step


def pmc_traffic_entry(path, key):
    """HBM bytes from a committed rocprofv3 --pmc pass (profiles/pmc_traffic*.json).  The entry is only used when it was collected on the
    kernel sources that are running now: it carries the hash of foldseek_amd/csrc/*.{hip,hpp,h} at collection time (tools/csrc_hash.py)."""
    try:
        e = json.load(open(path)).get(str(key))
    except Exception:
        return None, "no PMC file"
    if not e:
        return None, "no PMC entry for this DB size"
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import csrc_hash
        now = csrc_hash.csrc_hash(e.get("csrc_files"))
    except Exception:
        now = None
    if e.get("csrc_hash") != now or now is None:
        print(f"bench.py: WARNING: {os.path.basename(path)}[{key}] was collected on other kernel sources (hash {e.get('csrc_hash')} != {now}): traffic = null", file=sys.stderr)
        return None, f"stale: PMC pass belongs to kernel sources {e.get('csrc_hash')}, running {now}"
    return e, e.get("source")


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(relaunch_as_ranks(args))
    _claim_stdout()
    # host feeder threads spend their time inside the library (ctypes releases the GIL); a thread returning from a call
    # must not wait a whole default switch interval (5 ms) for the GIL while another one runs a few Python lines
    sys.setswitchinterval(1e-4)
    emu = max(0, args.emulate_rank_share)
    cores_before = usable_cores()
    if emu > 1:
        # what one rank of an `emu`-rank node has: its share of the cores this job may use (cgroup quota), pinned
        share = max(1, cores_before // emu)
        os.sched_setaffinity(0, set(sorted(os.sched_getaffinity(0))[:share]))
        os.environ["OMP_NUM_THREADS"] = str(share)
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
            if torch.cuda.device_count() < int(os.environ.get("LOCAL_WORLD_SIZE", str(world))):
                raise SystemExit(f"bench.py: {world} ranks over RCCL need one GPU each, {torch.cuda.device_count()} visible (FSGPU_BENCH_BACKEND=gloo shares devices for a plumbing check)")
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
    # One GPU: the DB "broadcast" still goes through a (one-rank) RCCL communicator, so that the transport the N-GPU run depends on has run and the
    # line says so (`broadcast_backend: nccl, rccl_ranks: 1`).  Default since round 5; FSGPU_REQUIRE_RCCL=1 makes a failure fatal, =0 skips it.
    rq = os.environ.get("FSGPU_REQUIRE_RCCL", "")
    require_rccl = rq not in ("", "0")
    rccl_note = None
    if world == 1 and rq != "0" and have_gpu and not dist.is_initialized() and not args.dry_run:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
        try:
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", dev_index))
        except Exception as e:                                             # noqa: BLE001
            if require_rccl:
                raise
            rccl_note = "one-rank RCCL communicator not available: " + repr(e)[:300]
    if have_gpu:
        torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index) if have_gpu else torch.device("cpu")
    local_rank = dev_index

    if args.workload == "allvsall" and not args.dry_run:
        from foldseek_amd import api
        out = allvsall_run(args, api, synth, fdist, dev, rank, world, local_rank, args.targets, args.steps, args.warmup, with_cpu=(world == 1 and not args.no_cpu_baseline))
        if rank == 0:
            emit_line(out)
        if world > 1:
            dist.destroy_process_group()
        return
    nthreads = max(1, args.host_threads)
    G = max(1, args.group)
    # ---- queries: weak = every rank its own steps; strong = the same steps*G queries split contiguously over the ranks ----
    share_world = emu if (emu > 1 and args.scaling == "strong") else world          # --emulate-rank-share: this process is rank 0 of `emu`
    per_rank_timed = args.steps * G if args.scaling == "weak" else None
    n_timed_total = args.steps * G * (world if args.scaling == "weak" else 1)
    n_warm = args.warmup * G
    q_lo, q_hi = (int(x) for x in args.query_len.split(","))
    all_q3, all_qa = synth.make_queries(n_timed_total + world * n_warm, seed=1000, lo=q_lo, hi=q_hi)   # default 250..450: around the mean length 350; same on every rank
    # querylen_full_range leg: queries over the DB's own length range (30..2000), their homologs planted like everybody's
    n_full = max(0, args.fullrange_steps) * G if (q_lo, q_hi) != (30, 2000) and not args.dry_run else 0
    fr_q3, fr_qa = synth.make_queries(n_full * world, seed=3000, lo=30, hi=2000) if n_full else ([], [])
    phase = {}
    t_phase = [time.perf_counter()]

    def mark(name):
        now = time.perf_counter()
        phase[name] = phase.get(name, 0.0) + now - t_phase[0]
        t_phase[0] = now
    if args.scaling == "weak":
        t_lo, t_hi = rank * per_rank_timed, (rank + 1) * per_rank_timed
    else:
        t_lo, t_hi = fdist.shard_range(n_timed_total, rank, share_world)
    w_lo = n_timed_total + rank * n_warm
    q3 = all_q3[w_lo:w_lo + n_warm] + all_q3[t_lo:t_hi]            # this rank: warm-up queries first, then its timed shard
    qa = all_qa[w_lo:w_lo + n_warm] + all_qa[t_lo:t_hi]
    nq = len(q3)
    n_mine = t_hi - t_lo
    # a rank with few queries (strong scaling: 1000 queries / 8 ranks = 125) takes smaller steps, so that every feeder thread has a batch
    # and the fill / drain of the pipeline (first scan before any SW, last SW after the last scan) is a batch of 16-32, not of 64
    G_eff = G
    while args.scaling == "strong" and n_mine < (nthreads + 1) * G_eff and G_eff > 16:
        G_eff //= 2
    # ---- target DB: generated on rank 0 (vectorised), ONE broadcast (RCCL over xGMI), then resident in every GPU's HBM ----
    t_gen = time.perf_counter()
    db = synth.make_db_fast(args.targets, (all_q3 + fr_q3, all_qa + fr_qa), seed=20260923, homologs_per_query=args.homologs) if rank == 0 else None
    t_gen = time.perf_counter() - t_gen
    mark("db_generation")
    if have_gpu:
        torch.cuda.synchronize()
    tb = time.perf_counter()
    tensors, db = fdist.broadcast_db(db, dev)
    if have_gpu:
        torch.cuda.synchronize()
    if world == 1 and dist.is_initialized():
        try:
            for x in tensors:                  # the one-rank communicator: every buffer of the DB through ncclBroadcast once
                dist.broadcast(x, 0)
            torch.cuda.synchronize()
        except Exception as e:                                             # noqa: BLE001
            if require_rccl:
                raise
            rccl_note = "one-rank RCCL broadcast failed: " + repr(e)[:300]
            dist.destroy_process_group()
    t_bcast = time.perf_counter() - tb if (world > 1 or dist.is_initialized()) else 0.0
    mark("db_broadcast")
    bcast_backend = dist.get_backend() if dist.is_initialized() else "none"
    if world > 1 and have_gpu and backend == "nccl":
        # the one collective of the path must have gone over RCCL with one rank per device -- no silent fallback
        devs = fdist.gather_objects((rank, torch.cuda.current_device(), str(tensors[0].device)))
        if rank == 0:
            assert bcast_backend == "nccl" and len({d[1] for d in devs}) == world and all(d[2].startswith("cuda") for d in devs), (bcast_backend, devs)
    if args.dry_run:
        # everything up to here is the multi-rank plumbing; check what the other ranks received and stop
        import zlib
        digest = zlib.crc32(db.data3di.tobytes()[:1 << 20]) ^ zlib.crc32(np.ascontiguousarray(db.lengths).tobytes())
        sizes = fdist.gather_objects((rank, n_mine, int(db.n), int(digest)))
        if world > 1:
            dist.barrier()
        if rank == 0:
            emit_line({"metric": "residues aligned/sec (prefilter+align)", "value": 0.0, "unit": "residues/s", "n_gpus": world,
                              "steps": args.steps, "warmup": args.warmup, "dry_run": True, "scaling": args.scaling,
                              "backend": bcast_backend, "db_broadcast_s": t_bcast, "db_generation_s": t_gen, "queries_per_step_effective": G_eff,
                              "phase_wall_s": dict(phase, total_since_start=time.perf_counter() - T_START),
                              "ranks": [{"rank": r, "timed_queries": m, "db_entries": n, "db_digest": d} for r, m, n, d in sizes],
                              "config": {"workload": "dry run: no device work", "targets": int(db.n), "queries_per_step": G}})
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
    # ... and what the library divides when it picks the host or the device block aligner for the backtraces (<= 4 cores per GPU: device): the quota and
    # the affinity mask a rank sees are the whole job's
    if local_world > 1 and "FSGPU_CORES_PER_GPU" not in os.environ:
        os.environ["FSGPU_CORES_PER_GPU"] = str(max(1, usable_cores() // local_world))
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

    def step1(t, i, ss=None):
        ss = ss or searches
        hits = ss[t].prefilter(q3[i])
        res = ss[t].align(qa[i], q3[i], hits["id"])
        return hits, res

    # the rank's queries are processed in length order (the order is free; the reference sorts its GPU database by length for
    # the same reason): a batch then spans few register classes, i.e. few scan launches
    timed = sorted(range(n_warm, nq), key=lambda i: len(q3[i]))
    batches = [timed[k:k + G_eff] for k in range(0, len(timed), G_eff)]
    warm = [list(range(k, min(k + G, n_warm))) for k in range(0, n_warm, G)]
    mark("device_open")
    dt, rec, step = search_region(api, ctxs, searches, q3, qa, warm, batches, world, dev, fdist, True, nq)
    mark("main_region_with_warmup")
    kms, sms, counts, host = rec["kms"], rec["sms"], rec["counts"], rec["host"]
    # dominant-kernel duration on an otherwise idle GPU (in the timed region the feeder threads' launches overlap, which
    # stretches each kernel's wall time)
    solo_g, solo_s = [], []
    for i in range(n_warm, min(nq, n_warm + 8)):
        step1(0, i)
        solo_g.append(ctxs[0].kernel_ms(0)); solo_s.append(ctxs[0].kernel_ms(1))
    # one whole batch alone on the device, three times: the SW passes' solo figure is their fastest repetition (a single one moved by +-10 % from run to run)
    solo_swp = None
    _sw_ms = lambda p: sum(float(p[d][0]) for d in (0, 1) if p[d][0] >= 0)
    for _ in range(3):
        step(0, batches[len(batches) // 2])
        p_ = ctxs[0].sw_last_passes()
        if solo_swp is None or _sw_ms(p_) < _sw_ms(solo_swp):
            solo_swp = p_
    tot = fdist.gather_objects((counts[0], counts[1], n_mine, host))
    ctx = ctx0

    def gapless_roofline(kms_, lq_list, q_lo_solo=None):
        """HBM figures the contract asks for + the VALU issue roofline that actually binds the scan"""
        residues = db.residues
        VALU_PEAK = 1024 * 64 * (4.0 / 3.0) / 4.3 * 2.4
        n_launch = sum(k[1] for k in kms_)
        n_batched = sum(k[2] for k in kms_)
        scan_s = sum(k[0] for k in kms_) * 1e-3
        kreg = scan_s / max(1, n_launch)                   # average duration of ONE scan launch
        q_per_launch = n_batched / max(1, n_launch)
        cells_reg = float(np.mean(lq_list)) * residues * q_per_launch     # DP cells of one launch
        alg_q = residues + db.n                            # per query: every target residue read once (1 B) + 1 score byte written
        alg_bytes = alg_q * q_per_launch                   # per launch
        e, src = pmc_traffic_entry(os.path.join(ROOT, "profiles", "pmc_traffic.json"), args.targets)
        traffic = None if e is None else (e["fetch_correction"] * e["fetch_size_kb"] * 1024 + e["write_size_kb"] * 1024) * q_per_launch
        return {"bound": "hbm", "achieved": alg_bytes / kreg / 1e9, "peak": 8000.0, "unit": "GB/s",
                "frac": alg_bytes / kreg / 1e9 / 8000.0, "traffic": traffic, "traffic_source": src, "algorithmic_bytes": alg_bytes,
                "kernel": "k_gapless", "kernel_ms": kreg * 1e3, "queries_per_launch": q_per_launch, "launches": n_launch,
                "note": "the scan is VALU/LDS bound (Lq cell updates per target byte), see valu below and DESIGN.md",
                # 0.75 packed VALU lane-ops per DP cell (2 x v_pk_add_f16 clamp + 1 x v_pk_maximum3_f16 per 4 cells); these
                # issue once per 4.3 cycles per SIMD (measured, profiles/r01_valu_lds_issue_rate_ubench.txt):
                # 1024 SIMDs x 64 lanes x 4/3 cells / 4.3 cyc x 2.4 GHz
                # frac_ideal_4cyc: the same against an ideal 4.0-cycle issue of the packed instructions (the 4.3 is this repository's own measurement)
                "valu": {"achieved_gcups": cells_reg / kreg / 1e9, "peak_gcups": VALU_PEAK, "frac": cells_reg / kreg / 1e9 / VALU_PEAK,
                         "peak_gcups_ideal_4cyc": VALU_PEAK * 4.3 / 4.0, "frac_ideal_4cyc": cells_reg / kreg / 1e9 / (VALU_PEAK * 4.3 / 4.0)}}

    out = None
    t_pref_cpu = None
    if rank == 0:
        residues = db.residues
        nq_total = sum(x[2] for x in tot)
        nh, nr = sum(x[0] for x in tot), sum(x[1] for x in tot)
        value = nq_total * residues / dt
        VALU_PEAK = 1024 * 64 * (4.0 / 3.0) / 4.3 * 2.4
        kavg = float(np.mean(solo_g)) * 1e-3
        lq_timed = [len(q3[i]) for i in timed]
        solo_lq = float(np.mean([len(q3[i]) for i in range(n_warm, min(nq, n_warm + 8))]))
        alg_q = residues + db.n
        mean_lq = float(np.mean(lq_timed))
        cells = solo_lq * residues
        mine = tot[0][3]
        rf = gapless_roofline(kms, lq_timed)
        rf["valu"].update({"device_level_gcups": nq_total / world * mean_lq * residues / dt / 1e9,
                           "device_level_frac": nq_total / world * mean_lq * residues / dt / 1e9 / VALU_PEAK,
                           "device_level_frac_ideal_4cyc": nq_total / world * mean_lq * residues / dt / 1e9 / (VALU_PEAK * 4.3 / 4.0),
                           "note": "per launch = one multi-query k_gapless launch (scan batches of the feeder threads run one at a time, SW / selection kernels of the other threads co-run); device_level = cells of all timed queries of one rank / wall time"})
        rf["solo"] = {"note": "one single-query launch on an idle device", "kernel_ms": kavg * 1e3, "achieved": alg_q / kavg / 1e9, "frac": alg_q / kavg / 1e9 / 8000.0,
                      "valu_achieved_gcups": cells / kavg / 1e9, "valu_frac": cells / kavg / 1e9 / VALU_PEAK}
        out = {
            "metric": "residues aligned/sec (prefilter+align)",
            "value": value, "unit": "residues/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f16 (integer-exact, scaled 2^-11) gapless scan + i16 SW", "data": "synthetic",
            "config": {"workload": f"1 step = {G} queries (query lengths drawn from the DB's length model CLIPPED to {q_lo}..{q_hi}"
                                   + ("" if (q_lo, q_hi) == (30, 2000) else "; the same search with queries over the DB's own 30..2000 range is the `querylen_full_range` leg")
                                   + f") vs {db.n}-structure synthetic 3Di DB (mean len {residues / db.n:.0f}, {args.homologs} planted homologs per query): "
                                   f"gapless prefilter (all targets; queries of one register class share one multi-query scan launch) + top-1000 per query, then one multi-query fwd/rev structure SW launch per pass "
                                   f"(--alignment-type {args.alignment_type}; forward over all pairs, reversed over the pairs that pass the forward gates) "
                                   f"+ host gates + block-aligner backtrace of every accepted hit; {nthreads} host feeder threads per GPU run their steps concurrently",
                       "targets": db.n, "db_residues": residues, "mean_query_len": mean_lq, "query_len_range": [q_lo, q_hi], "max_seqs": 1000, "homologs_per_query": args.homologs,
                       "queries_per_step": G, "queries_per_step_effective": G_eff, "queries_total": nq_total, "host_threads_per_gpu": nthreads, "host_backtrace_workers_per_gpu": api.host_workers(),
                       "usable_cores": usable_cores(), "parallelism": f"query-shard x{world} ({args.scaling}), DB replicated by one RCCL broadcast"},
            "queries_per_s": nq_total / dt, "ms_per_query": 1e3 * dt / (nq_total / world),
            "hits_per_query": nh / nq_total, "alignments_per_query": nr / nq_total,
            # rank 0's host-side accounting of the align leg (sums over its feeder threads, per query)
            "align_leg": {"reverse_pass_fraction": mine["rev_pairs"] / max(1, tot[0][0]),
                          "host_backtrace_ms_per_query": 1e3 * mine["backtrace_s"] / max(1, n_mine),
                          # accepted hits whose start position + CIGAR came from the device block aligner / all accepted hits (rank 0's feeders);
                          # the default is by the cores the process may use: <= 4 device, more host (profiles/r06_backtrace_modes.txt)
                          "backtraces_on_device": [int(mine["bt_device"]), int(mine["bt_all"])],
                          "host_gates_ms_per_query": 1e3 * mine["gates_s"] / max(1, n_mine),
                          "host_profiles_ms_per_query": 1e3 * mine["profiles_s"] / max(1, n_mine),
                          "sw_call_wall_ms_per_query": 1e3 * mine["sw_wait_s"] / max(1, n_mine),
                          "sw_kernels_ms_per_query": float(np.mean(sms)), "sw_kernel_ms_single_query_solo": float(np.mean(solo_s)),
                          "roofline": sw_roofline(rec["swp"], args.alignment_type == 2, solo_swp, args.targets)},
            # per-launch duration of the dominant kernel from HIP events on the library's stream, averaged over the launches
            # of the TIMED region (the host threads overlap their launches there, which stretches each one); the same
            # kernel alone on the device is reported under "solo"
            "roofline": rf,
            "db_broadcast_s": t_bcast, "db_generation_s": t_gen, "broadcast_backend": bcast_backend, "rccl_ranks": world if bcast_backend == "nccl" else 0,
        }
        if rccl_note:
            out["rccl_note"] = rccl_note
        if emu > 1:
            out["emulated_rank_share"] = {"ranks": emu, "cores_of_this_rank": usable_cores(), "cores_of_the_job": cores_before,
                                          "timed_queries_of_this_rank": n_mine, "projected_value_all_ranks": emu * value,
                                          "note": "ONE rank's share of an N-rank node run on one GPU: value is this rank's rate; with a replicated DB and no per-step collective the ranks do not interact"}
        if not args.no_cpu_baseline and world == 1:          # the CPU baselines are an N = 1 item (rank 0 has the host to itself)
            hits, _ = step1(0, n_warm)
            out["cpu_baseline"] = cpu_baseline(db, q3[n_warm], qa[n_warm], hits["id"], args.alignment_type, args.cpu_sample_targets,
                                               [q3[i] for i in range(n_warm + 1, min(nq, n_warm + max(1, args.cpu_sample_queries)))])
            t_pref_cpu = out["cpu_baseline"].get("prefilter_s_sample")
    mark("solo_probes_and_cpu_baseline")

    # ---- configs[3]: the same search with --alignment-type 2 (3Di + AA substitution scores, Gotoh affine gaps) on the resident DB ----
    if args.type2_steps > 0 and args.alignment_type != 2:
        try:
            par2 = api.default_params()
            par2.alignmentType = 2
            searches2 = [api.Search(c, par2) for c in ctxs]
            nb2 = min(len(batches), max(nthreads, args.type2_steps))
            pick = np.linspace(0, len(batches) - 1, nb2 + nthreads).astype(np.int64)          # spread over the length-sorted batches
            sel2 = [batches[i] for i in pick]
            dt2, rec2, step2 = search_region(api, ctxs, searches2, q3, qa, sel2[:nthreads], sel2[nthreads:], world, dev, fdist, False, nq)
            solo2 = None
            for _ in range(3):
                step2(0, sel2[nthreads + (len(sel2) - nthreads) // 2])
                p_ = ctxs[0].sw_last_passes()
                if solo2 is None or _sw_ms(p_) < _sw_ms(solo2):
                    solo2 = p_
            n2 = sum(len(b) for b in sel2[nthreads:])
            tot2 = fdist.gather_objects((rec2["counts"][0], rec2["counts"][1], n2))
            if rank == 0:
                n2t = sum(x[2] for x in tot2)
                lq2 = [len(q3[i]) for b in sel2[nthreads:] for i in b]
                leg = {"workload": f"configs[3]: {len(sel2) - nthreads} more steps of {G_eff} queries on the same resident {db.n}-structure DB with --alignment-type 2 "
                                   f"(3Di + AA scores at aaFactor 1.4, affine gaps 10/1): same gapless prefilter, k_sw2 with two LDS tables, host gates, backtraces",
                       "value": n2t * db.residues / dt2, "unit": "residues/s", "steps": len(sel2) - nthreads, "ms_per_step": 1e3 * dt2 / max(1, len(sel2) - nthreads),
                       "queries_per_s": n2t / dt2, "ms_per_query": 1e3 * dt2 / max(1, n2t / world), "mean_query_len": float(np.mean(lq2)),
                       "hits_per_query": sum(x[0] for x in tot2) / max(1, n2t), "alignments_per_query": sum(x[1] for x in tot2) / max(1, n2t),
                       "sw_kernels_ms_per_query": float(np.mean(rec2["sms"])),
                       "backtraces_on_device": [int(rec2["host"]["bt_device"]), int(rec2["host"]["bt_all"])],
                       "roofline": gapless_roofline(rec2["kms"], lq2), "align_roofline": sw_roofline(rec2["swp"], True, solo2, args.targets)}
                if not args.no_cpu_baseline and world == 1:
                    hits, _ = step1(0, n_warm, searches2)
                    leg["cpu_baseline"] = cpu_baseline(db, q3[n_warm], qa[n_warm], hits["id"], 2, args.cpu_sample_targets, reuse_prefilter=t_pref_cpu)
                out["align_type2"] = leg
            for x in searches2:
                x.close()
        except Exception as e:                                             # noqa: BLE001 -- see the k-mer leg below
            if world > 1:
                raise
            if rank == 0:
                out["align_type2"] = {"error": repr(e)[:500]}

    mark("align_type2_leg")
    # ---- the DB's own query length range (30..2000): short queries pair up in a scan kernel, queries beyond 896 residues run row-tiled ----
    if n_full > 0:
        try:
            f3, fa = fr_q3[rank * n_full:(rank + 1) * n_full], fr_qa[rank * n_full:(rank + 1) * n_full]
            order = sorted(range(n_full), key=lambda i: len(f3[i]))
            fb = [order[k:k + G_eff] for k in range(0, n_full, G_eff)]
            pickw = [fb[i] for i in np.linspace(0, len(fb) - 1, min(nthreads, len(fb))).astype(np.int64)]      # untimed: one batch per feeder thread, spread over the lengths
            dtf, recf, _ = search_region(api, ctxs, searches, f3, fa, pickw, fb, world, dev, fdist, True, n_full)
            totf = fdist.gather_objects((recf["counts"][0], recf["counts"][1], n_full))
            if rank == 0:
                nft = sum(x[2] for x in totf)
                lqf = [len(x) for x in f3]
                out["querylen_full_range"] = {
                    "workload": f"{len(fb)} more steps of {G_eff} queries with lengths drawn from the DB's own model over its whole range 30..2000 (mean {float(np.mean(lqf)):.0f}; "
                                f"{sum(1 for x in lqf if x > 896)} of {len(lqf)} beyond 896 residues = row-tiled scans, {sum(1 for x in lqf if x <= 256)} up to 256 = paired scans), "
                                f"same resident DB, --alignment-type {args.alignment_type}, {args.homologs} planted homologs per query",
                    "value": nft * db.residues / dtf, "unit": "residues/s", "steps": len(fb), "ms_per_step": 1e3 * dtf / max(1, len(fb)),
                    "queries_per_s": nft / dtf, "ms_per_query": 1e3 * dtf / max(1, nft / world), "mean_query_len": float(np.mean(lqf)),
                    "hits_per_query": sum(x[0] for x in totf) / max(1, nft), "alignments_per_query": sum(x[1] for x in totf) / max(1, nft),
                    "roofline": gapless_roofline(recf["kms"], lqf), "align_roofline": sw_roofline(recf["swp"], args.alignment_type == 2)}
        except Exception as e:                                             # noqa: BLE001 -- see the k-mer leg below
            if world > 1:
                raise
            if rank == 0:
                out["querylen_full_range"] = {"error": repr(e)[:500]}
    mark("querylen_full_range_leg")

    for x in searches[1:]:
        x.close()
    for c in ctxs[1:]:
        c.close()
    searches, ctxs = searches[:1], ctxs[:1]
    kout = None
    if not args.no_kmer:
        # the legs after the main line are additional evidence: on one GPU a failure in one of them is recorded in its place instead of
        # costing the whole line (with several ranks the others would wait in a collective: there the error propagates)
        try:
            kout = kmer_section(args, api, synth, ctx0, searches[0], par, db, rank, world, dev, fdist, [q3[i] for i in timed], [qa[i] for i in timed])
        except Exception as e:                                             # noqa: BLE001
            if world > 1:
                raise
            kout = {"error": repr(e)[:500]}
    if rank == 0 and kout is not None:
        if not args.no_cpu_baseline and world == 1 and "error" not in kout:
            kout["cpu_baseline"] = kmer_cpu_baseline(args, synth, db)
        out["kmer_prefilter"] = kout
    mark("kmer_leg")
    for x in searches:
        x.close()
    for c in ctxs:
        c.close()
    del db
    # ---- configs[1]: one query at a time against a 100k-structure DB (latency), rank 0's GPU only ----
    if args.single_targets > 0 and rank == 0:
        try:
            out["single_query_100k"] = single_query_leg(args, api, synth, local_rank, with_cpu=(world == 1 and not args.no_cpu_baseline))
        except Exception as e:                                             # noqa: BLE001
            out["single_query_100k"] = {"error": repr(e)[:500]}
    mark("single_query_leg")
    # ---- configs[4]: all-vs-all of a 200k-structure DB (easy-cluster's prefilter + align step), own DB, same process ----
    if args.allvsall_steps > 0:
        try:
            av = allvsall_run(args, api, synth, fdist, dev, rank, world, local_rank, args.allvsall_targets, args.allvsall_steps, 8,
                              with_cpu=(world == 1 and not args.no_cpu_baseline))
        except Exception as e:                                             # noqa: BLE001
            if world > 1:
                raise
            av = {"error": repr(e)[:500]}
        if rank == 0:
            for k in ("metric", "higher_is_better", "vs_baseline", "data"):
                av.pop(k, None)
            out["allvsall"] = av
    mark("allvsall_leg")
    if rank == 0:
        out["phase_wall_s"] = dict(phase, total_since_start=time.perf_counter() - T_START)
        emit_line(out)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
