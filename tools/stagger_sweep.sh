#!/bin/bash
# sweep of the staggered-start knob (LOLHIP_STAGGER=units,log2div,first/256)
run() { echo -n "[$LOLHIP_STAGGER] "; tools/bench_kernels "$@" | tail -1; }
for qb in 30 26; do
  unset LOLHIP_STAGGER; run 14 1 4096 polymul 300 $qb
  for d in 8 3; do for u in 1 2 4 6; do export LOLHIP_STAGGER=$u,$d,4; run 14 1 4096 polymul 300 $qb; done; done
done
unset LOLHIP_STAGGER; run 14 1 4096 crt 300 30
for u in 1 2 4; do export LOLHIP_STAGGER=$u,8,4; run 14 1 4096 crt 300 30; done
unset LOLHIP_STAGGER; run 14 1 4096 polymul 300 60
for d in 8 3; do for u in 2 4 8; do export LOLHIP_STAGGER=$u,$d,2; run 14 1 4096 polymul 300 60; done; done
