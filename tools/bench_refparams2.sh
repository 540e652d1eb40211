#!/bin/bash
# the reference's own index shapes (m = 2^e * odd): fused one-launch route vs the split route (LOLHIP_NO_FUSED2=1)
for m in 1728 5184 14400 11648 14336 96; do
  for qb in 26 58; do
    for op in crt polymul; do echo -n "fused: "; tools/bench_kernels m$m 1 8192 $op 50 $qb | tail -1; echo -n "split: "; LOLHIP_NO_FUSED2=1 tools/bench_kernels m$m 1 8192 $op 50 $qb | tail -1; done
  done
done
