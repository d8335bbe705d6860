#!/bin/bash
# sweep: gapless workgroups per CU (0 = adaptive default) x host threads
run() { python bench.py --no-cpu-baseline --no-kmer "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-40s ms/step %.3f  solo gapless %.3f  overlapped %.3f sw/q %.3f' % ('perCU=$FSGPU_GAPLESS_BLOCKS_PER_CU $*', d['ms_per_step'], d['roofline']['solo']['kernel_ms'], d['roofline']['kernel_ms'], d['roofline']['sw_kernels_ms_per_query']))"; }
for rep in 1 2; do
  unset FSGPU_GAPLESS_BLOCKS_PER_CU; run --host-threads 3
  for b in 2 3; do export FSGPU_GAPLESS_BLOCKS_PER_CU=$b; run --host-threads 3; done
done
unset FSGPU_GAPLESS_BLOCKS_PER_CU; run --host-threads 2;  run --host-threads 4; run --host-threads 1
