"""diagnostic: which polynomials / coefficients of the pipelined poly-mul differ from the one-per-workgroup kernel"""
import sys, numpy as np, torch
sys.path.insert(0, ".")
import lol_amd
from oracle import lolmath as lm
for L in (12, 13):
    m = 2 ** (L + 1)
    q = lm.first_good_q(m, 2 ** 20)
    P = lol_amd.Plan([(2, L + 1)], [q])
    n = 1 << L
    for B in (7, 768, 1100, 2048, 4096):
        g = torch.Generator(device="cuda"); g.manual_seed(B)
        a = torch.randint(0, q, (B, n, 1), dtype=torch.int64, device="cuda", generator=g)
        b = torch.randint(0, q, (B, n, 1), dtype=torch.int64, device="cuda", generator=g)
        lol_amd.debug_set("NO_PIPE", True); ref = torch.empty_like(a); P.polymul(a, b, out=ref); lol_amd.debug_set("NO_PIPE", False)
        for rep in range(3):
            lol_amd.debug_set("FORCE_PIPE", True); out = torch.empty_like(a); P.polymul(a, b, out=out); lol_amd.debug_set("FORCE_PIPE", False)
            torch.cuda.synchronize()
            bad = (out != ref).view(B, n)
            rows = bad.any(dim=1).nonzero().flatten().tolist()
            msg = f"L={L} B={B} rep={rep}: {len(rows)} bad polys"
            if rows:
                r = rows[0]
                cols = bad[r].nonzero().flatten().tolist()
                msg += f"; first {rows[:8]}; poly {r}: {len(cols)} bad coeffs, first {cols[:6]} last {cols[-3:]}; got {out[r, cols[:4], 0].tolist()} want {ref[r, cols[:4], 0].tolist()}"
            print(msg, flush=True)
