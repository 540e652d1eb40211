#!/bin/bash
# adjacent small factors as ONE Kronecker stage (3 (x) 5 = an 8-vector, BIG kernels), same box and library:
# default plans against plans built with LOLHIP_NO_KRON=1
cd "$(dirname "$0")/.."
run() { tools/bench_kernels "$@" | tail -1 | sed 's/ algorithmic.*//'; }
for spec in "m15015 1024 29" "m15015 1024 26" "m1155 8192 26" "m105 65536 20"; do
  set -- $spec
  for op in crt polymul; do
    echo "3(x)5 merged: $(run $1 1 $2 $op 40 $3)"
    echo "separate    : $(LOLHIP_NO_KRON=1 run $1 1 $2 $op 40 $3)"
  done
done
