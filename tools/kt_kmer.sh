cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_prof; mkdir -p $O
rm -rf /tmp/kkt && rocprofv3 --kernel-trace --stats -d /tmp/kkt -o kt -- python $R/tools/kmer_bench.py 1000000 32 3 > /tmp/kkt.log 2>&1
python $R/tools/rocprof_summary.py $(find /tmp/kkt -name "*.db" | head -1) > $O/$1_kernel_trace_kmer_batch32_1M.txt 2>&1
grep -h "^COUNTS\|^rep\|^partition" /tmp/kkt.log >> $O/$1_kernel_trace_kmer_batch32_1M.txt
grep -v "rocprim\|k_db_\|k_kmer_mask\|extract\|unique\|compact\|rows3\|bitmap" $O/$1_kernel_trace_kmer_batch32_1M.txt | head -40
