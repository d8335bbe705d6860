#!/bin/bash
# A/B of the column-segment work list (FSGPU_GAPLESS_NOSPLIT=1 disables it) at several DB sizes
run() { python bench.py --no-cpu-baseline --no-kmer "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-46s ms/step %.3f  solo gapless %.3f' % ('nosplit=${FSGPU_GAPLESS_NOSPLIT:-0} $*', d['ms_per_step'], d['roofline']['solo']['kernel_ms']))"; }
for t in 5000 20000 100000; do
  for rep in 1 2; do
    unset FSGPU_GAPLESS_NOSPLIT; run --targets $t --steps 480
    export FSGPU_GAPLESS_NOSPLIT=1; run --targets $t --steps 480
  done
done
