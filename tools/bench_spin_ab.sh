#!/bin/bash
# host wait policy A/B: polling window of syncStream (FSGPU_SPIN_US) vs throughput and CPU time of the bench process
TIMEFORMAT="    cpu: %U user %S sys %R wall"
run() { time (python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('spin_us=${FSGPU_SPIN_US:-default} $*  ms/step %.3f  kmer ms/query %.3f' % (d['ms_per_step'], d['kmer_prefilter']['ms_per_query']))"); }
for s in 1000000 40 0; do export FSGPU_SPIN_US=$s; run --host-threads 3; run --host-threads 3; done
export FSGPU_SPIN_US=40; run --host-threads 2; run --host-threads 4; run --host-threads 1
