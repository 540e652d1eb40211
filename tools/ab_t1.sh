#!/bin/bash
# A/B on one box: 8-byte (LOLHIP_NO_T1=1) vs 16-byte-per-lane global I/O of the m = 2^k kernels, and the copy yardstick
cd "$(dirname "$0")/.."
for rep in 1 2; do
for op in copy0 copy1; do tools/bench_kernels 14 1 4096 $op 50 29 | tail -1; done
for qb in 26 29 30 60; do for op in crt crtinv polymul mul; do
  echo -n "t1:    "; tools/bench_kernels 14 1 4096 $op 50 $qb | tail -1
  echo -n "no_t1: "; LOLHIP_NO_T1=1 tools/bench_kernels 14 1 4096 $op 50 $qb | tail -1
done; done
done
for lm in 11 12 13 15; do for op in crt polymul; do
  echo -n "t1:    "; tools/bench_kernels $lm 1 8192 $op 50 29 | tail -1
  echo -n "no_t1: "; LOLHIP_NO_T1=1 tools/bench_kernels $lm 1 8192 $op 50 29 | tail -1
done; done
