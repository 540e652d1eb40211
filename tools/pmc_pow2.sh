#!/bin/bash
# usage: tools/pmc_pow2.sh <tag> <kernel name pattern> -- <bench_kernels args...>   (environment passes through)
# SQ counters of one kernel in separate rocprofv3 --pmc passes (no tracing flags beside them), summarised per dispatch.
R=${GRAFT_REPO_ROOT:-/root/repo}; tag=$1; pat=$2; shift 3
O=$R/gpurun_out/pmc_$tag; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_BUSY_CYCLES" \
           "SQC_ICACHE_MISSES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VMEM" \
           "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $O/p$i -- $R/tools/bench_kernels "$@" > $O/p$i.log 2>&1 || echo "pass $i failed (see $O/p$i.log)"
  python3 $R/tools/pmc_summary.py "$pat" $O/p$i
done
