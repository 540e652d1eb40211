#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): bench line, rocprofv3 kernel stats of the same command,
# HBM-traffic counters in separate passes, SQ counters, config 4 (mixed radix) trace + counters,
# pipeline bench, instruction-cost microbenchmarks.  Everything lands in gpurun_out/refresh/;
# tools/collect_profiles.py <tag> then condenses it into profiles/.
# Needs tools/bench_kernels, tools/microbench_ops, tools/microbench_bfly2 (make -C tools; the
# binaries travel with the gpurun snapshot).  rocprofv3 always gets the program itself after `--`.
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/refresh
rm -rf $O; mkdir -p $O
make -s -C $R/tools || true      # the development binaries are built, never committed
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 50 --warmup 5 > $O/bench.json 2> $O/bench.err
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-secondary > $O/bench_prof.json 2> $O/bench_prof.err
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
# ---- config 4: m = 15015, batch 1024, q ~ 2^30 and ~ 2^60 ---------------------------------------
for qb in 30 60; do for op in crt polymul; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/c4_kt_${op}_$qb -- $R/tools/bench_kernels m15015 1 1024 $op 50 $qb > $O/c4_${op}_$qb.txt 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $O/c4_pmc_${op}_${qb}_$c -- $R/tools/bench_kernels m15015 1 1024 $op 5 $qb > /dev/null 2>&1
  done
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O/c4_sq_${op}_$qb -- $R/tools/bench_kernels m15015 1 1024 $op 5 $qb > /dev/null 2>&1
done; done
echo "config 4 done"
# ---- the reference's own index shapes (m = 2^e * odd): fused one-launch route vs the split route ----
(cd $R && tools/bench_refparams2.sh) > $O/refparams.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ref_kt -- $R/tools/bench_kernels m14400 1 8192 polymul 50 26 > $O/ref_kt.txt 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $O/ref_sq -- $R/tools/bench_kernels m14400 1 8192 polymul 5 26 > /dev/null 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/ref_pmc_$c -- $R/tools/bench_kernels m14400 1 8192 polymul 5 26 > /dev/null 2>&1
done
echo "reference shapes done"
python3 $R/tools/bench_pipelines.py > $O/pipelines.jsonl 2> $O/pipelines.err
$R/tools/microbench_ops > $O/microbench_ops.txt 2>&1
$R/tools/microbench_bfly2 > $O/microbench_bfly2.txt 2>&1
echo "all done"
cat $O/bench.json
