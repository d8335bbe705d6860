#!/bin/bash
run() { python bench.py --no-cpu-baseline --no-kmer "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-40s ms/step %.3f  solo gapless %.3f  overlapped %.3f' % ('$*', d['ms_per_step'], d['roofline']['solo']['kernel_ms'], d['roofline']['kernel_ms']))"; }
run --steps 96 --warmup 24
run --steps 192 --warmup 24
run --steps 384 --warmup 24
run --steps 768 --warmup 24
run --steps 1536 --warmup 24
run --steps 192 --warmup 24
