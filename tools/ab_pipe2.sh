#!/bin/bash
# A/B on one box: pipelined poly-mul with 2q in a VGPR (main) vs in an SGPR (build/ab_pipe_q2s) vs the one-polynomial-per-workgroup kernel
cd "$(dirname "$0")/.."
for rep in 1 2 3; do for qb in 26 29 30; do
  echo -n "pipe:     "; tools/bench_kernels 14 1 4096 polymul 50 $qb | tail -1
  echo -n "pipe_q2s: "; LD_LIBRARY_PATH=build/ab_pipe_q2s tools/bench_kernels 14 1 4096 polymul 50 $qb | tail -1
  echo -n "no_pipe:  "; LOLHIP_NO_PIPE=1 tools/bench_kernels 14 1 4096 polymul 50 $qb | tail -1
done; done
for qb in 26 29 30; do for op in crt crtinv; do tools/bench_kernels 14 1 4096 $op 50 $qb | tail -1; done; done
