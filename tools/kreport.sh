#!/bin/bash
# tools/kreport.sh <src.hip> <mangled-name regex> [extra hipcc flags]: registers, spills, LDS, occupancy and an
# opcode histogram of the matching kernels of one translation unit (compile only, no GPU).
src=$1; pat=$2; shift 2
cd "$(dirname "$0")/../lol_amd/csrc"
tmp=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -I../../include "$@" -S --cuda-device-only $src -o $tmp/k.s 2>/dev/null || { echo compile failed; exit 1; }
for k in $(grep -oE "^_Z[A-Za-z0-9_]+:" $tmp/k.s | tr -d : | grep -E "$pat"); do
  awk -v k="^$k:" '$0 ~ k {p=1} p; p&&/s_endpgm/{exit}' $tmp/k.s > $tmp/kk.s
  echo "== $k"
  grep -A40 "\.name: *$k\$" $tmp/k.s | grep -E "\.(vgpr_count|vgpr_spill_count|sgpr_count|sgpr_spill_count|group_segment_fixed_size|private_segment_fixed_size):" | tr -s ' ' | tr '\n' ' '; echo
  awk '{print $1}' $tmp/kk.s | grep -E "^(v_|s_|ds_|buffer_|global_|scratch_)" | sed -E 's/_e(32|64)$//' | sort | uniq -c | sort -rn | awk '{printf "%s:%s ", $2, $1} END {print ""}' | fold -w 200
  echo "  valu=$(grep -cE '^\s+v_' $tmp/kk.s) salu=$(grep -cE '^\s+s_' $tmp/kk.s) s_nop=$(grep -cE '^\s+s_nop' $tmp/kk.s) ds=$(grep -cE '^\s+ds_' $tmp/kk.s) vmem=$(grep -cE '^\s+(buffer_|global_)' $tmp/kk.s) scratch=$(grep -cE '^\s+scratch_' $tmp/kk.s) waitcnt=$(grep -cE 's_waitcnt' $tmp/kk.s) readlane=$(grep -cE 'v_(read|write)lane' $tmp/kk.s)"
done
rm -rf $tmp
