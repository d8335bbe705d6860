mkdir -p gpurun_out/r3l
export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/foldseek_amd
W=/tmp/es1; rm -rf $W; mkdir -p $W; cd $W
EX=$GRAFT_REPO_ROOT/tests/golden/example_structures
$GRAFT_REPO_ROOT/oracle/_ref_full/bin/foldseek-fsgpu easy-search $EX/d1asha_ $EX gpu.m8 tmp_gpu --threads 2 -v 3 --gpu 1 > $GRAFT_REPO_ROOT/gpurun_out/r3l/easy_gpu.log 2>&1; echo rc=$?
ls -la tmp_gpu/*/ >> $GRAFT_REPO_ROOT/gpurun_out/r3l/easy_gpu.log 2>&1
tail -40 $GRAFT_REPO_ROOT/gpurun_out/r3l/easy_gpu.log
