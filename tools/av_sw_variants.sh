#!/bin/bash
# all-vs-all leg alone (configs[4], 24 batches of 1024 queries) under different settings of the batch SW: tools/av_sw_variants.sh "ENV=V ENV2=V" "..." ...
for v in "$@"; do echo "== $v"; env $v python bench.py --workload allvsall --targets 200000 --steps 24 --warmup 8 --no-cpu-baseline --allvsall-batch 1024 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['align_roofline']; print('qps',round(d['queries_per_s']),'corun',{k:round(r[k],3) for k in ('frac','kernel_ms_per_pass_pair')},'solo',{k:round(r['solo'][k],3) for k in ('frac','kernel_ms')})
print({k:round(v,2) for k,v in d.get('host_wall_ms_per_batch').items()})
"; done
