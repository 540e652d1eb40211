#!/bin/bash
# assemble every .hip translation unit (device side only) and audit it for the store-data hazard
cd "$(dirname "$0")/../lol_amd/csrc"
tmp=$(mktemp -d); rc=0
[ $# -eq 0 ] && set -- *.hip
for f in "$@"; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -I../../include -S --cuda-device-only $f -o $tmp/$(basename $f .hip).s 2>/dev/null ) &
  while [ $(jobs -r | wc -l) -ge 8 ]; do sleep 0.5; done
done
wait
python3 ../../tools/check_store_hazard.py $tmp/*.s | tail -${TAIL:-40} ; rc=${PIPESTATUS[0]}
rm -rf $tmp; exit $rc
