#!/bin/bash
# all-vs-all k_sw3 and k-mer batch solo figures on the batch the DEFAULT run measures (short main leg + all-vs-all leg), under env settings: tools/ava_solo_isolation.sh "ENV=V" ...
for v in "$@"; do env $v python bench.py --no-cpu-baseline --steps 2 --warmup 1 --type2-steps 0 --fullrange-steps 0 --no-kmer --single-targets 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
a=d['allvsall']; r=a['align_roofline']; k=a['roofline']['solo']; print('$v: all-vs-all k_sw3 solo %.3f ms frac %.3f | co-running %.2f ms | %d queries/s | k-mer batch solo %.3f ms (count %.3f lists %.3f, k_kmer_lists %.3f) frac %.4f' % (r['solo']['kernel_ms'], r['solo']['frac'], r['kernel_ms_per_pass_pair'], a['queries_per_s'], k['kernel_ms'], k['stage_ms']['count'], k['stage_ms']['lists'], k['stage_ms']['k_kmer_lists'], k['frac']))"; done
