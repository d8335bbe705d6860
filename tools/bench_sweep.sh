#!/bin/bash
# usage: bench_sweep.sh  -> ms_per_step for (threads, group) combinations of the main bench path
for cfg in "3 0" "3 1" "3 4" "3 8" "3 16" "2 8" "2 16" "4 8" "3 0" "3 8"; do
  set -- $cfg
  python bench.py --no-cpu-baseline --no-kmer --host-threads $1 --group $2 --steps 96 --warmup 8 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('threads $1 group $2: ms/step %.3f  gapless overlapped %.3f  sw/q overlapped %.3f' % (d['ms_per_step'], d['roofline']['kernel_ms_overlapped'], d['roofline']['sw_kernel_ms_overlapped']))"
done
