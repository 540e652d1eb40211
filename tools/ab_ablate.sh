#!/bin/bash
# ablations of the pipelined poly-mul (results are garbage; timing only): what each ingredient costs
cd "$(dirname "$0")/.."
for qb in 26 29; do
  echo -n "full:     "; tools/bench_kernels 14 1 4096 polymul 30 $qb | tail -1
  for v in NO_XPOSE NO_BFLY NO_TW NO_IO VALUONLY; do
    printf "%-9s " $v:; LD_LIBRARY_PATH=build/ab_pipe_$v tools/bench_kernels 14 1 4096 polymul 30 $qb | tail -1
  done
done
