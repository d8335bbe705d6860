#!/bin/bash
# separate rocprofv3 --pmc passes over tools/prof_kernels.py (one resident DB, a few scans + SW batches of one query)
cd /tmp && export TMPDIR=/tmp
R=/root/repo
run() { rocprofv3 --pmc "$@" -d /tmp/pmc_$1 -o p --output-format csv -- python $R/tools/prof_kernels.py --reps 2 > /tmp/pmc_$1.log 2>&1; }
run SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY
run SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
run FETCH_SIZE
run WRITE_SIZE
python $R/tools/pmc_summary.py /tmp/pmc_SQ_WAVES /tmp/pmc_SQ_LDS_BANK_CONFLICT /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE
