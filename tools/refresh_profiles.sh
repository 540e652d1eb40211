#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): bench line, rocprofv3 kernel stats of the same command,
# HBM-traffic counters in separate passes, pipeline bench.  Everything lands in gpurun_out/;
# tools/collect_profiles.py then condenses it into profiles/.
set -e
# needs tools/bench_kernels (make -C tools bench_kernels; the binary travels with the gpurun snapshot)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/refresh
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 50 --warmup 5 > $O/bench.json 2> $O/bench.err
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_prof.json 2> $O/bench_prof.err
echo "kernel trace done"
for op in crt polymul; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $O/pmc_${op}_$c -- $R/tools/bench_kernels 14 1 4096 $op 5 > /dev/null 2>&1
  done
done
echo "pmc traffic done"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS --output-format csv -d $O/pmc_sq1 -- $R/tools/bench_kernels 14 1 4096 polymul 5 > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_WAVES SQC_ICACHE_MISSES SQ_INST_LEVEL_VMEM --output-format csv -d $O/pmc_sq2 -- $R/tools/bench_kernels 14 1 4096 polymul 5 > /dev/null 2>&1
echo "pmc sq done"
python3 $R/tools/bench_pipelines.py > $O/pipelines.jsonl 2> $O/pipelines.err
python3 $R/tools/pmc_summary.py k_pow2 $O/pmc_crt_FETCH_SIZE $O/pmc_crt_WRITE_SIZE > $O/pmc_crt.txt
python3 $R/tools/pmc_summary.py k_pow2 $O/pmc_polymul_FETCH_SIZE $O/pmc_polymul_WRITE_SIZE $O/pmc_sq1 $O/pmc_sq2 > $O/pmc_polymul.txt
cat $O/bench.json; cat $O/pmc_crt.txt $O/pmc_polymul.txt
