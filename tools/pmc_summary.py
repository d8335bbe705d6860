#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output (counter_collection.csv) per kernel: mean counter value per dispatch.
usage: pmc_summary.py <dir-or-csv> ..."""
import csv, glob, os, sys, collections
for arg in sys.argv[1:]:
    files = [arg] if arg.endswith(".csv") else sorted(glob.glob(os.path.join(arg, "**", "*counter_collection.csv"), recursive=True))
    for f in files:
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        print(f"# {f}")
        for k, v in agg.items():
            if k.startswith("__amd") or "k_db_" in k:
                continue
            n = len(next(iter(v.values())))
            print(f"{k[:70]:70s} dispatches={n}")
            for c, x in sorted(v.items()):
                print(f"    {c:28s} mean/dispatch = {sum(x)/len(x):16.1f}")
