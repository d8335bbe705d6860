#!/bin/bash
# all-vs-all k_sw3 solo figure on the batch the DEFAULT run measures (short main leg + all-vs-all leg), under env settings: tools/ava_solo_isolation.sh "ENV=V" ...
for v in "$@"; do env $v python bench.py --no-cpu-baseline --steps 2 --warmup 1 --type2-steps 0 --fullrange-steps 0 --no-kmer --single-targets 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
a=d['allvsall']; r=a['align_roofline']; print('$v: all-vs-all k_sw3 solo %.3f ms frac %.3f | co-running %.2f ms | %d queries/s' % (r['solo']['kernel_ms'], r['solo']['frac'], r['kernel_ms_per_pass_pair'], a['queries_per_s']))"; done
