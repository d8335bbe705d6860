cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_prof; mkdir -p $O
rm -rf /tmp/skt && rocprofv3 --kernel-trace --stats -d /tmp/skt -o kt -- python $R/tools/sw2_probe.py 1024 8 > /tmp/skt.log 2>&1
python $R/tools/rocprof_summary.py $(find /tmp/skt -name "*.db" | head -1) > $O/kt_sw_probe_1024x8.txt 2>&1
grep "alignment-type" /tmp/skt.log | cut -c1-200 >> $O/kt_sw_probe_1024x8.txt
grep "k_sw3\|alignment-type\|^kernel" $O/kt_sw_probe_1024x8.txt | cut -c1-200
