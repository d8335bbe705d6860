#!/bin/bash
run() { python bench.py --no-cpu-baseline --no-kmer "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-34s ms/step %.3f  solo gapless %.3f  overlapped %.3f sw/q %.3f' % ('$FSGPU_GAPLESS_BLOCKS_PER_CU $*', d['ms_per_step'], d['roofline']['solo']['kernel_ms'], d['roofline']['kernel_ms'], d['roofline']['sw_kernels_ms_per_query']))"; }
for b in 1 2 3 4; do export FSGPU_GAPLESS_BLOCKS_PER_CU=$b; run --host-threads 3; run --host-threads 2; done
export FSGPU_GAPLESS_BLOCKS_PER_CU=2; run --host-threads 4; run --host-threads 3 --group 16; run --host-threads 4 --group 16
