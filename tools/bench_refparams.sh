#!/bin/bash
# the reference's own benchmark / tunnelling index sets (lol/Crypto/Lol/Benchmarks/Default.hs:42-50,
# lol-apps .../Benchmarks/Default.hs:49-56) through the C ABI: crt and fused poly-mul, batch 8192, q ~ 2^26 and 2^58
for m in 1728 5184 14400 11648 3640 5460 4095 2048 14336; do
  for qb in 26 58; do
    for op in crt polymul; do tools/bench_kernels m$m 1 8192 $op 50 $qb | tail -1; done
  done
done
