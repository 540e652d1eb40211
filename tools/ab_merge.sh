#!/bin/bash
# merged prime-power stages (plan.cpp merge_stages), class 2 of the vector interpreter: same box, three builds
#   new    : this library (18-/20-vectors instantiated)            merged programs / staged programs (LOLHIP_NO_MERGE)

#   old    : without the 18-/20-vector instantiations (-DLH_NO_VL: the round's earlier kernel), staged programs only
cd "$(dirname "$0")/.."
run() { tools/bench_kernels "$@" | tail -1 | sed 's/ algorithmic.*//'; }
for spec in "m14400 8192 26" "m14400 8192 29" "m1728 16384 26" "m225 65536 26" "m11648 8192 26" "m15015 1024 29"; do
  set -- $spec
  for op in crt polymul; do
    echo    "new merged : $(run $1 1 $2 $op 40 $3)"
    echo    "new staged : $(LOLHIP_NO_MERGE=1 run $1 1 $2 $op 40 $3)"

    echo    "old staged : $(LD_LIBRARY_PATH=build/ab_old LOLHIP_NO_MERGE=1 run $1 1 $2 $op 40 $3)"
  done
done
