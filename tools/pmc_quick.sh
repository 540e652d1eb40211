#!/bin/bash
# usage: tools/pmc_quick.sh [mNNN] [B] [qbits...]   per-wave instruction and wait counters of k_mixed for l / crt / polymul
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/pmcq; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
M=${1:-m15015}; B=${2:-1024}; shift 2; QB=${@:-30 60}
for qb in $QB; do for op in l crt polymul; do
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $O/${op}_$qb -- $R/tools/bench_kernels $M 1 $B $op 5 $qb > /dev/null 2>&1
echo "== $M B=$B $op $qb"; python3 $R/tools/pmc_summary.py k_mixed $O/${op}_$qb
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --output-format csv -d $O/${op}_${qb}_b -- $R/tools/bench_kernels $M 1 $B $op 5 $qb > /dev/null 2>&1
python3 $R/tools/pmc_summary.py k_mixed $O/${op}_${qb}_b
done; done
