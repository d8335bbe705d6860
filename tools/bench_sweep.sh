#!/bin/bash
# usage: bench_sweep.sh  -> ms_per_step for (threads, group) combinations of the main bench path
for cfg in "3 8" "4 8" "5 8" "6 8" "4 16" "6 16" "3 8" "4 4" "8 8"; do
  set -- $cfg
  python bench.py --no-cpu-baseline --no-kmer --host-threads $1 --group $2 --steps 192 --warmup 16 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('threads $1 group $2: ms/step %.3f  gapless overlapped %.3f  sw/q overlapped %.3f' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['sw_kernels_ms_per_query']))"
done
