#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of one k-mer batch of 32 queries at 1M targets per kernel family: r04_kmer_fetch.sh TAG  (env passes through)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04_prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
TAG=${1:-x}
kpass() { rm -rf /tmp/kpmc_$1; rocprofv3 --pmc "$@" -d /tmp/kpmc_$1 -o p --output-format csv -- python $R/tools/kmer_bench.py 1000000 32 1 > /tmp/kpmc_$1.log 2>&1; }
kpass FETCH_SIZE
kpass WRITE_SIZE
python $R/tools/pmc_family.py /tmp/kpmc_FETCH_SIZE /tmp/kpmc_WRITE_SIZE --from-first k_kmer_count --json $O/${TAG}_pmc_kmer_fetch_1M.json > $O/${TAG}_pmc_kmer_fetch_1M.txt 2>&1
grep -h "^COUNTS\|^rep\|^segments" /tmp/kpmc_FETCH_SIZE.log > $O/${TAG}_pmc_kmer_fetch_1M_counts.txt
python - $O/${TAG}_pmc_kmer_fetch_1M.json <<'P'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in sorted(d.items(), key=lambda kv: -kv[1]["counters"].get("FETCH_SIZE", {}).get("total", 0)):
    c = v["counters"]
    print("%-40s FETCH %12.0f KB  WRITE %12.0f KB" % (k[:40], c.get("FETCH_SIZE", {}).get("total", 0), c.get("WRITE_SIZE", {}).get("total", 0)))
P
