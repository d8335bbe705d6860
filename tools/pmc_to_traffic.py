#!/usr/bin/env python3
"""Turn the --pmc summaries of tools/r05_profile.sh (gpurun_out/r05_prof/TAG_*) into the per-unit HBM bytes bench.py reports as
`roofline.traffic` (profiles/pmc_traffic.json, profiles/pmc_traffic_kmer.json, profiles/pmc_traffic_allvsall.json), stamped with the hash of
the kernel sources they were collected on (tools/csrc_hash.py; bench.py drops an entry whose hash is not the running one).
usage: pmc_to_traffic.py gpurun_out/r05_prof TAG profiles/<name of the committed summary files' prefix>"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from csrc_hash import ROOT, csrc_hash

d, tag, prefix = sys.argv[1], sys.argv[2], sys.argv[3]
GAPLESS_FILES, KMER_FILES = ["k_gapless.hpp", "fs_kernels.h"], ["k_kmer.hpp", "fsgpu_kmer.hip", "fs_kernels.h"]
hash_g = open(os.path.join(d, f"{tag}_csrc_hash_gapless.txt")).read().strip()
hash_k = open(os.path.join(d, f"{tag}_csrc_hash_kmer.txt")).read().strip()
for name, box, files in (("scan", hash_g, GAPLESS_FILES), ("k-mer", hash_k, KMER_FILES)):
    if box != csrc_hash(files):
        print(f"WARNING: the {name} kernel sources changed since the pass was collected ({box} on the box, {csrc_hash(files)} here): bench.py will report traffic null", file=sys.stderr)

# --- main path: k_gapless per query slot
b = json.load(open(os.path.join(d, f"{tag}_pmc_bench_1M.json")))
g = b["fs::k_gapless"]["counters"]
pt = os.path.join(ROOT, "profiles", "pmc_traffic.json")
t = json.load(open(pt))
e = t["1000000"]
e.update({"fetch_size_kb": g["FETCH_SIZE"]["per_query"], "write_size_kb": g["WRITE_SIZE"]["per_query"], "fetch_correction": 2.0,
          "workload": f"1M-target synthetic DB (seed 20260923, 50 homologs per query), bench.py --steps 3 --warmup 1 --no-kmer --type2-steps 0 --allvsall-steps 0: "
                      f"{g['FETCH_SIZE']['dispatches']} launches covering {g['FETCH_SIZE']['queries']:.0f} query slots",
          "source": f"{prefix}_pmc_bench_1M_steps3.txt (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, per-query = counter total / (grid / (512 workgroups x 256 threads)); NOT collected in the run that prints it)",
          "csrc_hash": hash_g, "csrc_files": GAPLESS_FILES})
json.dump(t, open(pt, "w"), indent=1)
valu = g["SQ_INSTS_VALU"]["per_query"]
print(f"k_gapless: FETCH {e['fetch_size_kb']:.1f} KB x2 + WRITE {e['write_size_kb']:.1f} KB per query slot; SQ_INSTS_VALU per query slot {valu:.4e}")

# --- structure SW of the same run: k_sw3 (+ its image builder), per target pair.  Pairs of the pass = what the line of the SAME run reports
# (timed queries x hits per query x (1 + reversed fraction)) scaled by (steps + warm-up) / steps: the counters also cover the warm-up step; the eight
# single-query solo probes run k_sw and are not in the k_sw3 family.  An estimate to a few per cent, said so in the entry.
SW_FILES = ["k_sw3.hpp", "k_sw.hpp", "fsgpu_sw3.hip", "fs_kernels.h"]
hash_sw_box = open(os.path.join(d, f"{tag}_csrc_hash_sw.txt")).read().strip()


def sw_entry(counters_json, line_json, key, what, allvsall=False):
    """one entry of profiles/pmc_traffic_sw.json (key = '<targets>:<alignment type>') from a counter summary and the bench line of the SAME run"""
    if not (os.path.exists(counters_json) and os.path.exists(line_json)):
        return
    bb = json.load(open(counters_json))
    if "fs::k_sw3" not in bb:
        return
    line = json.load(open(line_json))
    sw = bb["fs::k_sw3"]["counters"]
    im = bb.get("fs::k_sw3_image", {}).get("counters", {})
    if allvsall:
        # pairs of every batch the counters saw: timed + warm-up + the five solo repetitions, each a forward + reversed pass pair
        at = "2"
        pairs = line["align_roofline"]["pairs_per_pass_pair"] * (line["steps"] + line["warmup"] + 5)
    else:
        at = line["config"]["workload"].split("--alignment-type ")[1][:1]
        nq = line["config"]["queries_total"]
        pairs = nq * line["hits_per_query"] * (1.0 + line["align_leg"]["reverse_pass_fraction"]) * (line["steps"] + line["warmup"] + 1) / line["steps"]
    fkb = sw["FETCH_SIZE"]["total"] + im.get("FETCH_SIZE", {}).get("total", 0.0)
    wkb = sw["WRITE_SIZE"]["total"] + im.get("WRITE_SIZE", {}).get("total", 0.0)
    ps = os.path.join(ROOT, "profiles", "pmc_traffic_sw.json")
    ts = json.load(open(ps)) if os.path.exists(ps) else {}
    ts[key] = {"kernel": "k_sw3 + k_sw3_image (batch structure SW, both passes)", "fetch_size_kb": fkb, "write_size_kb": wkb, "fetch_correction": 2.0,
               "pairs_estimate": pairs, "bytes_per_pair": (2.0 * fkb + wkb) * 1024.0 / max(pairs, 1.0), "alignment_type": at,
               "note": "pairs = what the pass's own bench line reports (queries x hits per query x (1 + reversed fraction), scaled to the batches the counters saw: "
                       "timed + warm-up + solo); hbm_bytes = 2 x FETCH_SIZE + WRITE_SIZE (wide reads: the LDS images and the target codes). Algorithmic bytes of a "
                       "pair: its target's codes (1 B per residue and table) + 16 B of result; the rest is the workgroups' LDS images (17-106 KB per workgroup of "
                       "8-64 pairs, mostly L2 hits). An estimate to a few per cent.",
               "source": f"{prefix}_{what} (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes; NOT collected in the run that prints it)",
               "csrc_hash": hash_sw_box, "csrc_files": SW_FILES}
    json.dump(ts, open(ps, "w"), indent=1)
    print(f"k_sw3 [{key}]: FETCH {fkb:.0f} KB x2 + WRITE {wkb:.0f} KB over ~{pairs:.0f} pairs = {ts[key]['bytes_per_pair']:.0f} B per pair")


if hash_sw_box != csrc_hash(SW_FILES):
    print(f"WARNING: the SW kernel sources changed since the pass was collected ({hash_sw_box} on the box, {csrc_hash(SW_FILES)} here): bench.py will report traffic null", file=sys.stderr)
sw_entry(os.path.join(d, f"{tag}_pmc_bench_1M.json"), os.path.join(d, f"{tag}_pmc_bench_1M_benchline.json"), "1000000:0", "pmc_bench_1M_steps3.txt")
sw_entry(os.path.join(d, f"{tag}_pmc_bench_1M_t2.json"), os.path.join(d, f"{tag}_pmc_bench_1M_t2_benchline.json"), "1000000:2", "pmc_bench_1M_t2_steps3.txt")
sw_entry(os.path.join(d, f"{tag}_pmc_allvsall_200k.json"), os.path.join(d, f"{tag}_pmc_allvsall_200k_benchline.json"), "200000:2", "pmc_allvsall_200k.txt", allvsall=True)

# --- k-mer prefilter: every kernel of one 32-query batch, per index hit
k = json.load(open(os.path.join(d, f"{tag}_pmc_kmer_1M.json")))
cnt = None
for line in open(os.path.join(d, f"{tag}_pmc_kmer_1M_counts.txt")):
    if line.startswith("COUNTS "):
        cnt = json.loads(line[7:])
hits, probes = cnt["index_hits"], cnt["similar_kmers"]
fam = {n: (v["counters"].get("FETCH_SIZE", {}).get("total", 0.0), v["counters"].get("WRITE_SIZE", {}).get("total", 0.0)) for n, v in k.items()}
# calibration on known byte counts in this pipeline's own access pattern.  Round 6 (stable partition of narrow records): k_kmer_scatter_stable
# reads every record + its coarse key once, coalesced (4 + 2 B per hit; its per-(tile, key) offsets are < 0.5 % of that); k_kmer_emit writes them
# once at the hit's stream position (4 + 2 B per hit; the (tile, key) counts beside them < 0.5 %).  Rounds 3-5 (8-byte records): k_kmer_bincount
# read 8 B per hit, k_kmer_scatter_coarse wrote 8 B per hit.
if "fs::k_kmer_scatter_stable" in fam:
    fcal = 6.0 * hits / (fam["fs::k_kmer_scatter_stable"][0] * 1024.0)
    wcal = 6.0 * hits / (fam["fs::k_kmer_emit"][1] * 1024.0)
    calnote = ("(4-byte records + 2-byte keys, one per lane): k_kmer_scatter_stable fetches exactly 6 B per hit, k_kmer_emit writes exactly 6 B per hit")
else:
    fcal = 8.0 * hits / (fam["fs::k_kmer_bincount"][0] * 1024.0)
    wcal = 8.0 * hits / (fam["fs::k_kmer_scatter_coarse"][1] * 1024.0)
    calnote = "(8-byte records, one per lane): k_kmer_bincount fetches exactly 8 B per hit, k_kmer_scatter_coarse writes exactly 8 B per hit"
rows, tot_raw, tot_cal = {}, 0.0, 0.0
for n, (f, w) in sorted(fam.items(), key=lambda kv: -(kv[1][0] + kv[1][1])):
    stream = not any(x in n for x in ("k_kmer_lists", "k_kmer_count"))          # those two are random 4-8 byte probes: raw counter
    cal = (f * (fcal if stream else 1.0) + w * wcal) * 1024.0
    rows[n.replace("fs::", "")] = {"fetch_kb": f, "write_kb": w, "bytes_calibrated": cal, "bytes_per_index_hit": cal / hits}
    tot_raw += (f + w) * 1024.0
    tot_cal += cal
pk = os.path.join(ROOT, "profiles", "pmc_traffic_kmer.json")
tk = json.load(open(pk))
lists = rows["k_kmer_lists"]
tk["1000000"] = {
    "kernel": "k_kmer_* + the scans of one prefilter batch",
    "workload": f"1M-target synthetic DB (seed 20260923), ONE batch of 32 queries (tools/kmer_bench.py 1000000 32 1): {probes:.0f} similar k-mers probed, {hits:.0f} index hits, {cnt['candidates']:.0f} double-diagonal candidates",
    "index_hits": hits, "probes": probes,
    "fetch_calibration": fcal, "write_calibration": wcal,
    "k_kmer_all_bytes_per_index_hit": tot_cal / hits, "k_kmer_all_bytes_per_index_hit_raw_counters": tot_raw / hits,
    "k_kmer_lists_bytes_per_probe": (lists["fetch_kb"] + lists["write_kb"] * wcal) * 1024.0 / probes,
    "per_kernel": rows,
    "note": "separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes. The counters are calibrated on known byte counts in this pipeline's own access pattern "
            + calnote + "; fetch_calibration / "
            "write_calibration are bytes per reported byte (MI355X_MICROARCH.md, HBM section: FETCH_SIZE reports half of a wide streaming read, other widths "
            "to be calibrated). The random 4-8 byte probes of k_kmer_count / k_kmer_lists keep the raw fetch counter.",
    "source": f"{prefix}_pmc_kmer_batch32_1M.txt (rocprofv3 --pmc passes of tools/kmer_bench.py 1000000 32 1; NOT collected in the run that prints it)",
    "csrc_hash": hash_k, "csrc_files": KMER_FILES}
json.dump(tk, open(pk, "w"), indent=1)
print(f"k-mer batch: {hits:.3e} hits, calibrations fetch {fcal:.3f} write {wcal:.3f}; {tot_cal / hits:.1f} B per index hit (raw counters {tot_raw / hits:.1f})")
for n, r in rows.items():
    if r["bytes_per_index_hit"] > 0.05:
        print(f"   {n:32s} {r['bytes_per_index_hit']:7.2f} B/hit  (fetch {r['fetch_kb']:.0f} KB, write {r['write_kb']:.0f} KB)")

# --- all-vs-all (configs[4]): every k_kmer_* kernel of a short run, per index hit, with the calibrations of the 1M batch above
pa = os.path.join(d, f"{tag}_pmc_allvsall_200k.json")
if os.path.exists(pa):
    a = json.load(open(pa))
    line = json.load(open(os.path.join(d, f"{tag}_pmc_allvsall_200k_benchline.json")))
    nq = line["queries_per_s"] * line["ms_per_step"] * 1e-3 * line["steps"]            # timed queries; the counters also cover the warm-up batches
    nb_all = line["steps"] + line["warmup"] + 5                                          # + the five solo repetitions (three until round 5)
    ahits = line["index_hits_per_query"] * line["queries_per_batch"] * nb_all
    tot = 0.0
    arows = {}
    for n, v in a.items():
        if "k_kmer" not in n and "rocprim" not in n:
            continue
        f, w = v["counters"].get("FETCH_SIZE", {}).get("total", 0.0), v["counters"].get("WRITE_SIZE", {}).get("total", 0.0)
        stream = not any(x in n for x in ("k_kmer_lists", "k_kmer_count"))
        cal = (f * (fcal if stream else 1.0) + w * wcal) * 1024.0
        arows[n.replace("fs::", "")] = {"fetch_kb": f, "write_kb": w, "bytes_calibrated": cal}
        tot += cal
    pav = os.path.join(ROOT, "profiles", "pmc_traffic_allvsall.json")
    tav = json.load(open(pav)) if os.path.exists(pav) else {}
    tav["200000"] = {"kernel": "k_kmer_* + the scans of the all-vs-all prefilter batches", "index_hits": ahits, "batches": nb_all,
                     "k_kmer_all_bytes_per_index_hit": tot / max(ahits, 1.0), "per_kernel": arows,
                     "workload": f"bench.py --workload allvsall --targets 200000 --steps {line['steps']} --warmup {line['warmup']} --kmer-threads 1 (family DB, batches of {line['queries_per_batch']})",
                     "note": "index hits = index_hits_per_query of the line x queries per batch x (timed + warm-up + five solo batches); fetch / write calibrations of the 1M k-mer batch pass",
                     "source": f"{prefix}_pmc_allvsall_200k.txt (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes; NOT collected in the run that prints it)",
                     "csrc_hash": hash_k, "csrc_files": KMER_FILES}
    json.dump(tav, open(pav, "w"), indent=1)
    print(f"all-vs-all: {ahits:.3e} index hits over {nb_all} batches, {tot / max(ahits, 1.0):.1f} B per index hit")
