#!/bin/bash
# round-5 A/B helper: r05_ab.sh TAG "ENV=1 ..." -- kmer_bench at 1M (3 reps of one 32-query batch) with the given environment, output under gpurun_out/r05/
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05; mkdir -p $O
TAG=$1; shift
env "$@" python $R/tools/kmer_bench.py 1000000 32 4 > $O/kb_$TAG.txt 2>&1
tail -3 $O/kb_$TAG.txt
