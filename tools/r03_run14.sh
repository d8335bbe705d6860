R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_prof
mkdir -p $O $R/gpurun_out/r3n
cd $R
for P in 8 16 32; do FSGPU_SW2_PAIRS=$P python tools/sw2_probe.py 2>&1 | grep alignment; done | tee gpurun_out/r3n/sw2_probe.txt
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/r3n/parity.log 2>&1; echo "parity rc=$?"; tail -3 gpurun_out/r3n/parity.log
cd /tmp && export TMPDIR=/tmp
TAG=r03_j
python $R/tools/csrc_hash.py k_kmer.hpp fsgpu_kmer.hip fs_kernels.h > $O/${TAG}_csrc_hash_kmer.txt
kpass() { rm -rf /tmp/kpmc_$1; rocprofv3 --pmc "$@" -d /tmp/kpmc_$1 -o p --output-format csv -- python $R/tools/kmer_bench.py 1000000 32 1 > /tmp/kpmc_$1.log 2>&1; }
kpass SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY
kpass SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
kpass FETCH_SIZE
kpass WRITE_SIZE
python $R/tools/pmc_family.py /tmp/kpmc_SQ_WAVES /tmp/kpmc_SQ_LDS_BANK_CONFLICT /tmp/kpmc_FETCH_SIZE /tmp/kpmc_WRITE_SIZE --from-first k_kmer_count --json $O/${TAG}_pmc_kmer_1M.json > $O/${TAG}_pmc_kmer_batch32_1M.txt 2>&1
grep -h "^COUNTS\|^rep\|^segments" /tmp/kpmc_FETCH_SIZE.log > $O/${TAG}_pmc_kmer_1M_counts.txt
