#!/bin/bash
# 61-bit fused poly-mul: what an s_nop and the forward butterfly's conditional subtraction cost
# (nop1/nop2: one more s_nop 0 / s_nop 1 per Shoup product; nocsub: results garbage, timing only)
cd "$(dirname "$0")/.."
make -s -C tools bench_kernels >/dev/null 2>&1
for r in 1 2; do
  echo -n "base:   "; tools/bench_kernels 14 1 4096 polymul 30 60 | tail -1
  for v in nop1 nop2 nocsub; do
    printf "%-7s " $v:; LD_LIBRARY_PATH=build/ab_$v tools/bench_kernels 14 1 4096 polymul 30 60 | tail -1
  done
done
