#!/bin/bash
# round-5 helper: r05_kt.sh TAG [ENV=1 ...] -- rocprofv3 kernel trace of three k-mer batches (32 queries, 1M targets), summary under gpurun_out/r05/
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05; mkdir -p $O
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kkt_$TAG && env "$@" rocprofv3 --kernel-trace --stats -d /tmp/kkt_$TAG -o kt -- python $R/tools/kmer_bench.py 1000000 32 3 > /tmp/kkt_$TAG.log 2>&1
python $R/tools/rocprof_summary.py $(find /tmp/kkt_$TAG -name "*.db" | head -1) > $O/kt_$TAG.txt 2>&1
grep -h "^COUNTS\|^rep\|^segments" /tmp/kkt_$TAG.log >> $O/kt_$TAG.txt
grep "^rep" /tmp/kkt_$TAG.log | tail -2
