#!/bin/bash
for kt in 1 2 3 4; do
  python bench.py --no-cpu-baseline --steps 48 --warmup 8 --kmer-threads $kt --kmer-queries 384 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kmer_prefilter']
print('kmer threads $kt: ms/query %.3f  q/s %.0f  prefilter wall %.3f align wall %.3f  sw/batch %.2f' % (k['ms_per_query'], k['queries_per_s'], k['prefilter_ms_per_query_host_wall'], k['align_ms_per_query_host_wall'], k['sw_kernels_ms_per_batch32']))"
done
