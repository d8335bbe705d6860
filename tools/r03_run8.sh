mkdir -p gpurun_out/r3h
timeout 900 python -m pytest tests/test_kmer_gpu.py -x -q > gpurun_out/r3h/kmer_gpu.log 2>&1; echo "kmer_gpu rc=$?"; tail -3 gpurun_out/r3h/kmer_gpu.log
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { rm -rf /tmp/pmc_$1; rocprofv3 --pmc "$@" -d /tmp/pmc_$1 -o p --output-format csv -- python $R/tools/kmer_bench.py 1000000 32 1 > /tmp/pmc_$1.log 2>&1; }
run SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY
run SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
grep -E "segments of|^rep" /tmp/pmc_SQ_WAVES.log
python $R/tools/pmc_family.py /tmp/pmc_SQ_WAVES /tmp/pmc_SQ_LDS_BANK_CONFLICT > $R/gpurun_out/r3h/pmc_kmer_bench.txt 2>&1
grep -A20 -E "k_kmer_(dup_wg|binscatter|bincount)" $R/gpurun_out/r3h/pmc_kmer_bench.txt | head -120
