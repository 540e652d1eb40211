#!/bin/bash
# usage: tools/ab_run.sh <variant dirs under build/ ...>: config 4 and reference shapes through each variant ("main" = lol_amd/liblolhip.so)
for v in main "$@"; do
  if [ $v = main ]; then unset LD_LIBRARY_PATH; else export LD_LIBRARY_PATH=build/$v; fi
  echo "== $v"
  tools/bench_c4.sh | grep "2^30"
  for m in 1728 14400 11648; do for op in crt polymul; do tools/bench_kernels m$m 1 8192 $op 50 26 | tail -1; done; done
done
