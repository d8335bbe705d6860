#!/bin/bash
# per-kernel PMC sums of the solo SW probe (tools/sw2_probe.py): tools/pmc_sw_probe.sh TAG "COUNTER ..." [ENV=VALUE ...]   (counters only: never combined with tracing)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06_prof; mkdir -p $O
TAG=$1; CNT=$2; shift 2
rm -rf /tmp/pk && env "$@" rocprofv3 --pmc $CNT -d /tmp/pk -o p --output-format csv -- python $R/tools/sw2_probe.py > /tmp/pk.log 2>&1
python3 - > $O/${TAG}_pmc_sw_probe.txt <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for path in glob.glob("/tmp/pk/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"][:60]
        if "k_sw3" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[(k, r["Counter_Name"])] += 1
for k, d in sorted(acc.items()):
    print(k)
    for c, v in sorted(d.items()): print(f"   {c:24s} {v:16.0f}   ({calls[(k, c)]} dispatches)")
PY
grep alignment-type /tmp/pk.log >> $O/${TAG}_pmc_sw_probe.txt
cat $O/${TAG}_pmc_sw_probe.txt
