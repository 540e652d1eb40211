#!/usr/bin/env python3
"""Crude VGPR liveness over a straight-line stretch of gfx950 assembly (hipcc -S output):
prints the live-register count along the stretch and where it peaks.  Usage:
  asm_pressure.py file.s first_line last_line"""
import re, sys
lines = open(sys.argv[1]).read().split("\n")
lo, hi = int(sys.argv[2]), int(sys.argv[3])
NODEF = ("buffer_store", "scratch_store", "ds_write", "global_store", "v_cmp", "s_", "v_nop", ";", "v_cmpx")
def regs(tok):
    out = []
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(1): out += list(range(int(m.group(1)), int(m.group(2)) + 1))
        else: out.append(int(m.group(3)))
    return out
ins = []
for i in range(lo - 1, hi):
    l = lines[i].split(";")[0].strip()
    if not l or l.startswith(".") or l.endswith(":"): continue
    op, _, rest = l.partition(" ")
    ops = [o.strip() for o in rest.split(",")]
    defs, uses = [], []
    if ops and not op.startswith(NODEF):
        defs = regs(ops[0]); srcs = ops[1:]
        # accumulate forms read their destination only when it is also listed as a source
    else:
        srcs = ops
    for o in srcs: uses += regs(o)
    ins.append((i + 1, op, set(defs), set(uses)))
live = set(); counts = []
for ln, op, d, u in reversed(ins):
    live -= d; live |= u
    counts.append((ln, len(live), op))
counts.reverse()
peak = max(c for _, c, _ in counts)
print("peak live VGPRs:", peak)
step = max(1, len(counts) // 60)
for k in range(0, len(counts), step):
    seg = counts[k:k + step]
    m = max(seg, key=lambda x: x[1])
    print(f"lines {seg[0][0]:6d}-{seg[-1][0]:6d}  max live {m[1]:4d} at {m[0]} ({m[2]})")
