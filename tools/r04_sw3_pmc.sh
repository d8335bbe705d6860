#!/bin/bash
# Round 4: counters and kernel trace of the batch SW probe (tools/sw2_probe.py: 32 queries x 1000 targets, forward pass, 3Di and 3Di + AA).
# usage: r04_sw3_pmc.sh TAG   (env such as FSGPU_SW3_LONG / FSGPU_SW_PROFILES is passed through)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04_prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
TAG=${1:-x}
pass() { rm -rf /tmp/pmc_$1; rocprofv3 --pmc "$@" -d /tmp/pmc_$1 -o p --output-format csv -- python $R/tools/sw2_probe.py > /tmp/pmc_$1.log 2>&1; }
pass SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY
pass SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
python $R/tools/pmc_family.py /tmp/pmc_SQ_WAVES /tmp/pmc_SQ_LDS_BANK_CONFLICT --json $O/${TAG}_pmc_sw_probe.json > $O/${TAG}_pmc_sw_probe.txt 2>&1
rm -rf /tmp/kt && rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $R/tools/sw2_probe.py > $O/${TAG}_probe_under_trace.txt 2>&1
python $R/tools/rocprof_summary.py $(find /tmp/kt -name "*.db" | head -1) > $O/${TAG}_kernel_trace_sw_probe.txt 2>&1
