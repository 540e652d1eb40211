#!/bin/bash
# Runs ON THE GPU BOX: kernel-trace statistics and FETCH_SIZE / WRITE_SIZE (separate --pmc passes, nothing else beside
# them) of tools/bench_pipelines.py — the pipeline and streaming kernels (k_keyswitch, k_knapsack, k_decompose, k_ctmul,
# k_gather*, k_twace_crt, k_coeffs, k_rescale, k_pointwise_mul*).  Output: gpurun_out/pipes/{kt,pmc_FETCH_SIZE,pmc_WRITE_SIZE};
# tools/summarize_pipelines_pmc.py condenses them into profiles/r03_pipelines_pmc.json.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/pipes; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/tools/bench_pipelines.py > $O/kt.jsonl 2> $O/kt.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- python3 $R/tools/bench_pipelines.py > $O/pmc_$c.jsonl 2> $O/pmc_$c.err
done
# calibration on a known byte count in the same access form: the 16-byte-per-lane copy of 256 MiB
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/cal_$c -- $R/tools/bench_kernels 14 1 4096 copy0 5 29 > /dev/null 2>&1
  rocprofv3 --pmc $c --output-format csv -d $O/hl_$c -- $R/tools/bench_kernels 14 1 4096 polymul 5 60 > /dev/null 2>&1
  rocprofv3 --pmc $c --output-format csv -d $O/pp_$c -- $R/tools/bench_kernels 14 1 4096 polymul 5 26 > /dev/null 2>&1
done
echo pipelines pmc done
