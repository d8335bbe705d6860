#!/bin/bash
# k-mer leg of bench.py under 1 .. 6 feeder threads: queries/s and how far a batch's device part stretches against its solo time
for n in "$@"; do python bench.py --steps 3 --warmup 1 --type2-steps 0 --allvsall-steps 0 --fullrange-steps 0 --single-targets 0 --no-cpu-baseline --kmer-threads $n 2>/dev/null | python -c "
import json,sys
k=json.loads(sys.stdin.read().strip().splitlines()[-1])['kmer_prefilter']
print('threads %d: %.0f queries/s, device part per query %.3f ms in the timed region / %.3f solo = %.2fx, host wall prefilter %.3f align %.3f ms/query' % (k['host_threads'], k['queries_per_s'], k['prefilter_device_ms_per_query'], k['prefilter_device_ms_per_query_solo'], k['prefilter_device_ms_per_query']/k['prefilter_device_ms_per_query_solo'], k['prefilter_ms_per_query_host_wall'], k['align_ms_per_query_host_wall']))
"; done
