#!/bin/bash
# per-kernel PMC sums of one k-mer batch run: tools/pmc_kernel.sh TAG "COUNTER ..." [kernel-substring]   (counters only: never combined with tracing)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06_prof; mkdir -p $O
rm -rf /tmp/pk && rocprofv3 --pmc $2 -d /tmp/pk -o p --output-format csv -- python $R/tools/kmer_bench.py 1000000 32 1 > /tmp/pk.log 2>&1
python3 - "$3" > $O/$1_pmc_kernel.txt <<'PY'
import csv, glob, sys, collections
sub = sys.argv[1] if len(sys.argv) > 1 else ""
f = glob.glob("/tmp/pk/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for path in f:
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"][:60]
        if sub and sub not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[(k, r["Counter_Name"])] += 1
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()): print(f"   {c:24s} {v:16.0f}   ({calls[(k, c)]} dispatches)")
PY
cat $O/$1_pmc_kernel.txt
