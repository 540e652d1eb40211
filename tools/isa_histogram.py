#!/usr/bin/env python3
"""tools/isa_histogram.py <csrc dir> [label] — per-wave opcode histogram of the fused poly-mul kernels
(k_pow2<13,2,AR>) from the gfx950 ISA hipcc emits for that source tree: multiply-class, other
VALU, s_nop, LDS, VMEM, SALU, spills.  Static counts (the kernels are straight-line: every
instruction is executed once per wave).  Used for profiles/r02_isa_histogram.txt:
    git worktree add /tmp/r01 8d77a3f && python3 tools/isa_histogram.py /tmp/r01/lol_amd/csrc round-1
    python3 tools/isa_histogram.py lol_amd/csrc round-2"""
import collections
import os
import re
import subprocess
import sys
import tempfile

src = sys.argv[1]
label = sys.argv[2] if len(sys.argv) > 2 else src
inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(src))), "include")
MUL = {"v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mul_u32_u24", "v_mad_u32_u24", "v_mad_i64_i32", "v_mul_hi_i32"}
print(f"== {label}")
for ar in (1, 2, 3, 4):
    with tempfile.NamedTemporaryFile(suffix=".s") as tmp:
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++20", f"-I{inc}", f"-I{src}", "-S", "--cuda-device-only",
                        os.path.join(src, f"pow2_ar{ar}.hip"), "-o", tmp.name], check=True, capture_output=True)
        text = open(tmp.name).read()
    m = re.search(rf"^(_ZN6lolhip6k_pow2ILi13ELi2ELi{ar}ELb0E[^:]*):[^\n]*\n(.*?)s_endpgm", text, re.S | re.M)
    body = m.group(2)
    ops = collections.Counter(l.split()[0] for l in body.splitlines() if re.match(r"^\s+[vsdb][a-z0-9_]+", l))
    valu = sum(c for o, c in ops.items() if o.startswith("v_"))
    mul = sum(c for o, c in ops.items() if o.split("_e")[0] in MUL or o in MUL)
    meta = re.search(rf"\.name:\s+{re.escape(m.group(1))}.*?\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)", text, re.S)
    print(f"k_pow2<13,2,{ar}>: VALU {valu} (multiply-class {mul}, other {valu - mul})  s_nop {ops.get('s_nop', 0)}  "
          f"LDS {sum(c for o, c in ops.items() if o.startswith('ds_'))}  VMEM {sum(c for o, c in ops.items() if o.startswith('buffer_') or o.startswith('global_') or o.startswith('scratch_'))}  "
          f"SALU+SMEM {sum(c for o, c in ops.items() if o.startswith('s_') and o not in ('s_nop', 's_waitcnt'))}  s_waitcnt {ops.get('s_waitcnt', 0)}  "
          f"VGPRs {meta.group(1) if meta else '?'}  spilled {meta.group(2) if meta else '?'}")
    top = ", ".join(f"{o} {c}" for o, c in ops.most_common(14))
    print(f"    top: {top}")
