#!/usr/bin/env python3
"""tools/check_store_hazard.py <file.s | file.o | lib.so ...> — static audit of gfx950 code for the store-data hazard
(assembly text, or the device code objects bundled in a host object / shared library, disassembled with llvm-objdump):
a VMEM store of more than 8 bytes followed, fewer than 2 wait states later, by a VALU write to one of its
data VGPRs.  hipcc (ROCm 7.2) pads this pair only when the buffer store's soffset is NOT an SGPR; on MI355X the
SGPR-soffset form corrupts as well (profiles/r03_store_hazard.txt: a persistent loop's last store picked up the
loop-bound compare's v_mov).  Exit status 1 if any unpadded pair is found."""
import re
import sys

# data operand: first for buffer_store, second (after the address) for global/flat/scratch stores
STORE = re.compile(r"^\s*(?:buffer_store_dwordx[34]\s+|(?:global|flat|scratch)_store_dwordx[34]\s+(?:v\[\d+:\d+\]|v\d+|off),\s*)(v\[(\d+):(\d+)\])")
DST = re.compile(r"^\s*v_\w+\s+(v(\d+)|v\[(\d+):(\d+)\])")
SKIP = re.compile(r"^\s*(;|\.|$)|^\S+.*:$")


def dst_regs(line):
    m = DST.match(line)
    if not m or line.lstrip().startswith(("v_cmp", "v_cmpx", "v_readlane", "v_readfirstlane")):
        return set()
    if m.group(2) is not None:
        return {int(m.group(2))}
    return set(range(int(m.group(3)), int(m.group(4)) + 1))


OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def device_disassembly(path):
    """every gfx950 code object of the clang offload bundles inside a host ELF, disassembled"""
    import struct
    import subprocess
    import tempfile
    data = open(path, "rb").read()
    magic, out, pos = b"__CLANG_OFFLOAD_BUNDLE__", [], 0
    while True:
        i = data.find(magic, pos)
        if i < 0:
            break
        pos = i + len(magic)
        n, = struct.unpack_from("<Q", data, i + 24)
        off = i + 32
        for _ in range(n):
            o, sz, tl = struct.unpack_from("<QQQ", data, off)
            off += 24
            triple = data[off:off + tl].decode(errors="replace")
            off += tl
            if "gfx950" in triple and sz:
                with tempfile.NamedTemporaryFile(suffix=".co") as f:
                    f.write(data[i + o:i + o + sz])
                    f.flush()
                    out += subprocess.run([OBJDUMP, "-d", f.name], capture_output=True, text=True, check=True).stdout.splitlines()
    if not out:
        raise SystemExit(f"{path}: no gfx950 code object found")
    return [ln.split("//")[0].rstrip() for ln in out]


def audit(path):
    lines = open(path).read().splitlines() if path.endswith(".s") else device_disassembly(path)
    kernel, bad = "?", []
    for i, ln in enumerate(lines):
        mk = re.match(r"^(?:[0-9a-f]+ <)?(_Z\w+)>?:", ln)
        if mk:
            kernel = mk.group(1)
        m = STORE.match(ln)
        if not m:
            continue
        data = set(range(int(m.group(2)), int(m.group(3)) + 1))
        states, j = 0, i + 1
        while j < len(lines) and states < 2:
            nx = lines[j]
            j += 1
            if SKIP.match(nx):
                continue
            if nx.lstrip().startswith(("s_endpgm", "s_branch", "s_cbranch", "s_setpc")):
                break                                   # control flow: not followed further (reported separately below)
            mn = re.match(r"^\s*s_nop\s+(\d+)", nx)
            if mn:
                states += int(mn.group(1)) + 1
                continue
            if dst_regs(nx) & data:
                bad.append((kernel, i + 1, ln.strip(), nx.strip(), states))
                break
            states += 1
    return bad


def main():
    total = 0
    for p in sys.argv[1:]:
        for kernel, line, st, nx, states in audit(p):
            total += 1
            print(f"{p}:{line}: {kernel}\n    {st}\n    {nx}    <- after {states} wait state(s)")
    print(f"{total} unpadded store-data pair(s)")
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
