#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes per kernel FAMILY (template arguments stripped): dispatches and counter totals, plus
derived per-query figures for the scan kernel (a multi-query k_gapless launch runs grid / (512 workgroups x 256 threads)
queries: 2 workgroups per CU x 256 CUs per query, fsgpu.hip::launchGapless; --wg-per-query N overrides).
usage: pmc_family.py <dir-with-counter_collection.csv> ... [--json out.json]"""
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
fam = collections.defaultdict(lambda: {"dispatches": collections.defaultdict(int), "sum": collections.defaultdict(float), "grid": collections.defaultdict(float)})
for d in args:
    for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            name = re.sub(r"<.*", "", r["Kernel_Name"].replace("void ", "")).split("(")[0]
            if name.startswith("__amd") or "k_db_" in name:
                continue
            c = r["Counter_Name"]
            e = fam[name]
            e["dispatches"][c] += 1
            e["sum"][c] += float(r["Counter_Value"])
            e["grid"][c] += float(r["Grid_Size"])
res = {}
for name, e in sorted(fam.items(), key=lambda kv: -max(kv[1]["sum"].values())):
    row = {"counters": {}}
    for c in sorted(e["sum"]):
        row["counters"][c] = {"dispatches": e["dispatches"][c], "total": e["sum"][c]}
        if "k_gapless" in name:
            queries = e["grid"][c] / (WG * 256)
            row["counters"][c]["queries"] = queries
            row["counters"][c]["per_query"] = e["sum"][c] / max(queries, 1e-9)
    res[name] = row
    print(name)
    for c, v in row["counters"].items():
        extra = f"  queries={v['queries']:.1f} per_query={v['per_query']:.1f}" if "per_query" in v else ""
        print(f"    {c:24s} dispatches={v['dispatches']:5d} total={v['total']:18.1f}{extra}")
if out_json:
    json.dump(res, open(out_json, "w"), indent=1)
