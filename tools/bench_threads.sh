#!/bin/bash
# host feeder threads x group size sweep (main path only)
run() { python bench.py --no-cpu-baseline --no-kmer "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-36s ms/step %.3f  solo gapless %.3f  sw/q %.3f' % ('$*', d['ms_per_step'], d['roofline']['solo']['kernel_ms'], d['roofline']['sw_kernels_ms_per_query']))"; }
for t in 3 4 5 6; do run --host-threads $t; done
run --host-threads 3 --group 32; run --host-threads 4 --group 32; run --host-threads 4 --group 8; run --host-threads 6 --group 8
