#!/bin/bash
# A/B on one box: the interpreter's own-diagonal fast path (one division per vector) vs two divisions per element (LOLHIP_NO_OWN_DIAG=1)
cd "$(dirname "$0")/.."
for rep in 1 2; do for m in 14400 11648 1728; do for op in crt polymul; do
  echo -n "own_diag: "; tools/bench_kernels m$m 1 8192 $op 50 26 | tail -1
  echo -n "general:  "; LOLHIP_NO_OWN_DIAG=1 tools/bench_kernels m$m 1 8192 $op 50 26 | tail -1
done; done; done
