#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/pmcq; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for qb in 30 60; do for op in l crt; do
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $O/${op}_$qb -- $R/tools/bench_kernels m15015 1 1024 $op 5 $qb > /dev/null 2>&1
echo "== $op $qb"; python3 $R/tools/pmc_summary.py k_mixed $O/${op}_$qb
done; done
