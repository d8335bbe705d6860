#!/bin/bash
# The batch SW next to the other feeders' scans under different device-sharing schemes: tools/corun_variants.sh "ENV=V ..." ...
# Each variant: main leg (configs[2], 20 steps of 64 queries) + the --alignment-type 2 leg (12 steps), nothing else.
for v in "$@"; do echo "== $v"; env $v python bench.py --no-cpu-baseline --no-kmer --allvsall-steps 0 --fullrange-steps 0 --single-targets 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['align_leg']['roofline']; t=d['align_type2']; r2=t['align_roofline']
print('main ms/step %.2f  k_sw3 co-run frac %.3f (%.2f ms per pass pair, solo %.3f / %.2f ms) | type2 ms/step %.2f  k_sw3 co-run frac %.3f (%.2f ms, solo %.3f / %.2f ms) | sw_call_wall ms/query %.3f' % (
  d['ms_per_step'], r['frac'], r['kernel_ms_per_pass_pair'], r['solo']['frac'], r['solo']['kernel_ms'], t['ms_per_step'], r2['frac'], r2['kernel_ms_per_pass_pair'], r2['solo']['frac'], r2['solo']['kernel_ms'], d['align_leg']['sw_call_wall_ms_per_query']))
"; done
