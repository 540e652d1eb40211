#!/bin/bash
# A/B on one box: one-polynomial-per-workgroup fused poly-mul (LOLHIP_NO_PIPE=1) vs the persistent DMA-pipelined kernel
cd "$(dirname "$0")/.."
for rep in 1 2 3; do
for qb in 26 29 30; do
  echo -n "pipe:    "; tools/bench_kernels 14 1 4096 polymul 50 $qb | tail -1
  echo -n "no_pipe: "; LOLHIP_NO_PIPE=1 tools/bench_kernels 14 1 4096 polymul 50 $qb | tail -1
done; done
for qb in 26 29; do
  echo -n "pipe:    "; tools/bench_kernels 13 1 8192 polymul 50 $qb | tail -1
  echo -n "no_pipe: "; LOLHIP_NO_PIPE=1 tools/bench_kernels 13 1 8192 polymul 50 $qb | tail -1
done
