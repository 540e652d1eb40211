#!/bin/bash
# Runs ON THE GPU BOX: the driver's own command, then the SAME command under rocprofv3 --kernel-trace --stats
# (no secondary legs / CPU baseline in the traced run so the statistics are the metric's kernel alone).
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/benchprof; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/bench_prof.json 2> $O/bench_prof.err
python3 $R/tools/summarize_rocprof.py $O/prof $O/kernel_stats.csv > /dev/null
head -4 $O/kernel_stats.csv; cat $O/bench_prof.json | cut -c1-400
