mkdir -p gpurun_out/r3m
export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/foldseek_amd
W=/tmp/es2; rm -rf $W; mkdir -p $W; cd $W
EX=$GRAFT_REPO_ROOT/tests/golden/example_structures
CPU=$GRAFT_REPO_ROOT/oracle/_ref_full/bin/foldseek; GPU=$GRAFT_REPO_ROOT/oracle/_ref_full/bin/foldseek-fsgpu
$CPU createdb $EX target --threads 2 -v 1; cp target_h target_h.saved; $CPU makepaddedseqdb target target_pad --threads 2 -v 1; cp target_h.saved target_h
COLS="--prefilter-mode 1 --format-output query,target,fident,alnlen,mismatch,gapopen,qstart,qend,tstart,tend,evalue,bits,cigar"
$CPU easy-search $EX target_pad cpu.m8 tmp_cpu --threads 2 -v 1 $COLS
$GPU easy-search $EX target_pad gpu.m8 tmp_gpu --threads 2 -v 1 --gpu 1 $COLS
cp cpu.m8 gpu.m8 $GRAFT_REPO_ROOT/gpurun_out/r3m/
diff cpu.m8 gpu.m8 | head -20
