#!/bin/bash
# config 4 (m = 15015, batch 1024) through the C ABI with HIP events, both moduli
for qb in 30 60; do for op in l crt crtinv polymul; do tools/bench_kernels m15015 1 1024 $op 200 $qb | tail -1; done; done
