cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_prof; mkdir -p $O
rm -rf /tmp/bkt && rocprofv3 --kernel-trace --stats -d /tmp/bkt -o kt -- python $R/tools/btrace_probe.py > /tmp/bkt.log 2>&1
python $R/tools/rocprof_summary.py $(find /tmp/bkt -name "*.db" | head -1) > $O/y_kernel_trace_btrace_probe.txt 2>&1
grep "^queries" /tmp/bkt.log >> $O/y_kernel_trace_btrace_probe.txt
grep "k_block\|^queries" $O/y_kernel_trace_btrace_probe.txt
