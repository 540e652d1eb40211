#!/bin/bash
# resource usage + spill/waitcnt proxies of the fused poly-mul kernels (compile only; AR = 1 translation unit)
cd /root/repo/lol_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -I/root/repo/include "$@" -S --cuda-device-only pow2_ar1.hip -o /root/repo/gpurun_out/k.s 2>/dev/null
for pat in "k_pow2ILi13ELi2ELi1ELb0E" "k_pow2ILi9ELi2ELi1ELb0E" "k_pow2ILi9ELi2ELi1ELb1E" "k_pow2ILi7ELi2ELi1ELb1E"; do
  awk -v pat="^_ZN6lolhip6$pat[^:]*:" '$0 ~ pat {p=1} p&&/s_endpgm/{p=0} p' /root/repo/gpurun_out/k.s > /root/repo/gpurun_out/kk.s
  printf "%s lines=%s scratch=%s vmcnt0=%s subbrev=%s " $pat $(wc -l < /root/repo/gpurun_out/kk.s) $(grep -c scratch_ /root/repo/gpurun_out/kk.s) $(grep -c "vmcnt(0)" /root/repo/gpurun_out/kk.s) $(grep -c subbrev /root/repo/gpurun_out/kk.s)
  grep -A12 "name:.*$pat" /root/repo/gpurun_out/k.s | grep -E "vgpr_count|vgpr_spill|sgpr_count" | tr -d '\n'; echo
done
