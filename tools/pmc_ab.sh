set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p /tmp/vo && cp $R/tools/var_old_liblolhip.so /tmp/vo/liblolhip.so
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  unset LD_LIBRARY_PATH
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmcab/new$i -- $R/tools/bench_kernels 14 1 4096 polymul 3 > /dev/null 2>&1
  export LD_LIBRARY_PATH=/tmp/vo
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmcab/old$i -- $R/tools/bench_kernels 14 1 4096 polymul 3 > /dev/null 2>&1
  echo "set $i done"
done
unset LD_LIBRARY_PATH
cd $R
for v in old new; do echo "== $v"; python3 tools/pmc_summary.py k_pow2 gpurun_out/pmcab/${v}1 gpurun_out/pmcab/${v}2 gpurun_out/pmcab/${v}3; done > gpurun_out/pmcab_summary.txt
