#!/bin/bash
# A/B helper: tools/ab_build.sh <name> "<extra -D flags>" <src.hip ...>
# Recompiles only the named translation units with the extra flags and links them with the
# current objects into build/ab_<name>/liblolhip.so, so that one GPU call can time both:
#   LD_LIBRARY_PATH=build/ab_<name> tools/bench_kernels ...   (bench_kernels uses RUNPATH)
set -e
name=$1; flags=$2; shift 2
cd "$(dirname "$0")/../lol_amd/csrc"
out=../../build/ab_$name; mkdir -p $out/obj
objs=""
for o in obj/*.o; do
  b=$(basename $o .o); hit=0
  for s in "$@"; do [ "$b" = "$(basename $s .hip)" ] && hit=1; done
  if [ $hit = 1 ]; then objs="$objs $out/obj/$b.o"; else objs="$objs $o"; fi
done
for s in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -fvisibility=hidden -I../../include $flags -c -o $out/obj/$(basename $s .hip).o $s &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/liblolhip.so $objs
echo built $out/liblolhip.so
