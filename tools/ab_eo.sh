#!/bin/bash
# even/odd form of the odd primes' dense stages (class 3: 64-bit residues), same box: this library vs the same
# library with the dense form (build/ab_noeo: tools/ab_build.sh noeo -DLH_NO_EO mixed_cls3.hip mixed_cls3f12.hip mixed_cls3f16.hip)
cd "$(dirname "$0")/.."
run() { tools/bench_kernels "$@" | tail -1 | sed 's/ algorithmic.*//'; }
for spec in "m15015 1024 60" "m15015 1024 45" "m11648 8192 58" "m14400 8192 58" "m5005 4096 58"; do
  set -- $spec
  for op in crt polymul; do
    echo "even/odd: $(run $1 1 $2 $op 30 $3)"
    echo "dense   : $(LD_LIBRARY_PATH=build/ab_noeo run $1 1 $2 $op 30 $3)"
  done
done
