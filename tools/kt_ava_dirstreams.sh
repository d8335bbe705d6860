cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_prof; mkdir -p $O
for v in 0 1; do
rm -rf /tmp/akt && FSGPU_SW3_DIRSTREAMS=$v rocprofv3 --kernel-trace --stats -d /tmp/akt -o kt -- python $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 --type2-steps 0 --fullrange-steps 0 --no-kmer --single-targets 0 --allvsall-steps 6 > /tmp/akt.log 2>&1
python $R/tools/rocprof_summary.py $(find /tmp/akt -name "*.db" | head -1) > $O/kt_ava_dir$v.txt 2>&1
echo "== DIRSTREAMS=$v"; grep "k_sw3<" $O/kt_ava_dir$v.txt | cut -c1-150
done
