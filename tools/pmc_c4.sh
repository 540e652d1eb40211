#!/bin/bash
# rocprofv3 counters + kernel trace for config 4 (m = 15015, batch 1024); output under gpurun_out/r2_c4
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2_c4; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for qb in 30 60; do for op in l crt polymul; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_${op}_$qb -- $R/tools/bench_kernels m15015 1 1024 $op 50 $qb > $O/kt_${op}_$qb.txt 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O/pmc1_${op}_$qb -- $R/tools/bench_kernels m15015 1 1024 $op 5 $qb > /dev/null 2>&1
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS SQC_ICACHE_MISSES SQ_INST_LEVEL_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA --output-format csv -d $O/pmc2_${op}_$qb -- $R/tools/bench_kernels m15015 1 1024 $op 5 $qb > /dev/null 2>&1
  echo "== $op q~2^$qb"; python3 $R/tools/pmc_summary.py k_mixed $O/pmc1_${op}_$qb $O/pmc2_${op}_$qb
  for f in $(find $O/kt_${op}_$qb -name "*kernel_stats.csv"); do grep -E "k_mixed|Name" $f | cut -c1-200; done
done; done
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_${c} -- $R/tools/bench_kernels m15015 1 1024 polymul 5 30 > /dev/null 2>&1
  python3 $R/tools/pmc_summary.py k_mixed $O/pmc_${c}
done
