#!/bin/bash
# A/B counters of the scan kernels: queries of 600..700 residues (R = 38..44, 8-wave workgroups) against 300..350 (R = 19..22).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
SHORT="--steps 3 --warmup 1 --no-cpu-baseline --no-kmer"
pass() { n=$1; shift; q=$1; shift; rm -rf /tmp/pmc_$n; rocprofv3 --pmc "$@" -d /tmp/pmc_$n -o p --output-format csv -- python $R/bench.py $SHORT --query-len $q > /tmp/pmc_$n.log 2>&1; }
for q in 600,700 300,350; do
  t=$(echo $q | tr , _)
  pass c_$t $q SQ_IFETCH SQ_WAIT_IFETCH SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD
  tail -3 /tmp/pmc_c_$t.log | cut -c1-300
  python $R/tools/pmc_family.py /tmp/pmc_c_$t 2>&1 | sed -n 1,9p
  echo ====
done
