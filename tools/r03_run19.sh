mkdir -p gpurun_out/r3x
W=/tmp/ec; rm -rf $W; mkdir -p $W; cd $W
EX=$GRAFT_REPO_ROOT/tests/golden/example_structures
FS=$GRAFT_REPO_ROOT/oracle/_ref_full/bin
export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/foldseek_amd:$LD_LIBRARY_PATH
$FS/foldseek easy-cluster $EX cpu tmp_cpu --threads 1 -v 1 > cpu.log 2>&1; echo "cpu rc=$?"
$FS/foldseek-fsgpu easy-cluster $EX gpu tmp_gpu --threads 1 -v 3 --gpu 1 > gpu.log 2>&1; echo "gpu rc=$?"
tail -3 gpu.log
grep -c "Index table (device)" gpu.log; grep -E "^(prefilter|structurealign) " gpu.log | grep -c -- "--gpu 1"
cmp cpu_cluster.tsv gpu_cluster.tsv && echo "cluster tsv identical"; wc -l cpu_cluster.tsv gpu_cluster.tsv
cp gpu.log $GRAFT_REPO_ROOT/gpurun_out/r3x/easy_cluster_gpu.log; cp cpu_cluster.tsv gpu_cluster.tsv $GRAFT_REPO_ROOT/gpurun_out/r3x/ 2>/dev/null
