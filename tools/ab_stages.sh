#!/bin/bash
# vector interpreter, class 2: what each stage of the crt program costs (timing-only builds that skip
# stage s of every program: build/ab_skip<mask>, tools/ab_build.sh skip<mask> -DLH_ABL_SKIP_STAGES=<mask> mixed_cls2.hip)
cd "$(dirname "$0")/.."
for m in m14400 m11648 m15015; do
  B=8192; [ $m = m15015 ] && B=1024
  qb=26; [ $m = m15015 ] && qb=29
  for op in crt polymul; do
    echo -n "$m $op full:    "; tools/bench_kernels $m 1 $B $op 30 $qb | tail -1
    for v in 1 2 4 8 16 32 63; do
      printf "$m $op skip%-3s " $v:; LD_LIBRARY_PATH=build/ab_skip$v tools/bench_kernels $m 1 $B $op 30 $qb | tail -1
    done
  done
done
