#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes per kernel FAMILY (template arguments stripped): dispatches and counter totals, plus
derived per-query figures for the scan kernel (a multi-query k_gapless launch runs grid / (512 workgroups x 256 threads)
queries: 2 workgroups per CU x 256 CUs per query, fsgpu.hip::launchGapless; --wg-per-query N overrides).
--from-first NAME drops every dispatch before the first one of kernel NAME (per pass): e.g. the index build in front of a k-mer batch.
usage: pmc_family.py <dir-with-counter_collection.csv> ... [--json out.json] [--from-first k_kmer_count]"""
import collections
import csv
import glob
import json
import os
import re
import sys

WG = int(sys.argv[sys.argv.index("--wg-per-query") + 1]) if "--wg-per-query" in sys.argv else 512
args = [a for a in sys.argv[1:] if not a.startswith("--") and not a.isdigit()]
out_json = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
if out_json in args:
    args.remove(out_json)
first = sys.argv[sys.argv.index("--from-first") + 1] if "--from-first" in sys.argv else None
if first in args:
    args.remove(first)
fam = collections.defaultdict(lambda: {"dispatches": collections.defaultdict(int), "sum": collections.defaultdict(float), "grid": collections.defaultdict(float),
                                       "ns": collections.defaultdict(float)})
for d in args:
    for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        rows = list(csv.DictReader(open(f)))
        start = min([int(r["Dispatch_Id"]) for r in rows if first and first in r["Kernel_Name"]] or [0])
        for r in rows:
            if int(r["Dispatch_Id"]) < start:
                continue
            name = re.sub(r"<.*", "", r["Kernel_Name"].replace("void ", "")).split("(")[0]
            if name.startswith("__amd") or "k_db_" in name:
                continue
            c = r["Counter_Name"]
            e = fam[name]
            e["dispatches"][c] += 1
            e["sum"][c] += float(r["Counter_Value"])
            e["grid"][c] += float(r["Grid_Size"])
            e["ns"][c] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
res = {}
for name, e in sorted(fam.items(), key=lambda kv: -max(kv[1]["sum"].values())):
    row = {"counters": {}}
    for c in sorted(e["sum"]):
        row["counters"][c] = {"dispatches": e["dispatches"][c], "total": e["sum"][c], "kernel_ms": e["ns"][c] * 1e-6}
        if "k_gapless" in name:
            queries = e["grid"][c] / (WG * 256)
            row["counters"][c]["queries"] = queries
            row["counters"][c]["per_query"] = e["sum"][c] / max(queries, 1e-9)
    if "SQ_INSTS_VALU" in row["counters"]:
        # issue-rate cross-check (the rooflines of k_gapless / k_sw2 price a packed VALU instruction at 4.3 cycles per SIMD): the
        # dispatches of this pass ran one at a time (counter collection serialises them), so wave-instructions / 1024 SIMDs over their
        # summed duration at 2.4 GHz is the cycles the kernel really spends per VALU instruction and SIMD; 4.3 / that = its fraction of the bound
        v = row["counters"]["SQ_INSTS_VALU"]
        cyc = v["kernel_ms"] * 1e-3 * 2.4e9 * 1024 / max(v["total"], 1.0)
        row["valu_cycles_per_instr_per_simd"] = cyc
        row["valu_issue_frac_at_4.3_cycles"] = 4.3 / cyc if cyc > 0 else None
    res[name] = row
    print(name)
    if "valu_cycles_per_instr_per_simd" in row:
        print(f"    SQ_INSTS_VALU cross-check: {row['valu_cycles_per_instr_per_simd']:.2f} cycles per VALU wave-instruction and SIMD over {row['counters']['SQ_INSTS_VALU']['kernel_ms']:.1f} ms "
              f"of serialised dispatches -> {row['valu_issue_frac_at_4.3_cycles']:.3f} of the 4.3-cycle issue bound")
    for c, v in row["counters"].items():
        extra = f"  queries={v['queries']:.1f} per_query={v['per_query']:.1f}" if "per_query" in v else ""
        print(f"    {c:24s} dispatches={v['dispatches']:5d} total={v['total']:18.1f}{extra}")
if out_json:
    json.dump(res, open(out_json, "w"), indent=1)
