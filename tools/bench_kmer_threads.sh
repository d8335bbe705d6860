#!/bin/bash
run() { python bench.py --no-cpu-baseline --steps 96 --warmup 6 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())['kmer_prefilter']
print('%-44s ms/query %.3f  prefilter dev %.3f  host wall pref %.3f align %.3f' % ('$*', d['ms_per_query'], d['prefilter_device_ms_per_query'], d['prefilter_ms_per_query_host_wall'], d['align_ms_per_query_host_wall']))"; }
for t in 3 4 6; do run --kmer-threads $t --kmer-queries 768; run --kmer-threads $t --kmer-queries 768; done
