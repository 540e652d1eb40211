#!/bin/bash
# class 4 of the vector interpreter (class 2 with lazy dense stages, q < 2^27), same box and library:
# default plans against plans built with LOLHIP_NO_LAZY=1 (class 2)
cd "$(dirname "$0")/.."
run() { tools/bench_kernels "$@" | tail -1 | sed 's/ algorithmic.*//'; }
for spec in "m14400 8192 26" "m11648 8192 26" "m15015 1024 26" "m1728 16384 20" "m225 65536 20"; do
  set -- $spec
  for op in crt polymul; do
    echo "lazy (4): $(run $1 1 $2 $op 40 $3)"
    echo "class 2 : $(LOLHIP_NO_LAZY=1 run $1 1 $2 $op 40 $3)"
  done
done
