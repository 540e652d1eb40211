"""Timing of generic-path operations through the Python binding (development aid)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import lol_amd
from oracle import lolmath as lm
def run(m, lower, B, T=1, ops=("crt","crtInv","l","mulGPow","divGDec","polymul")):
    pps = lm.factor_pps(m); g = lm.good_qs(m, lower); qs=[next(g) for _ in range(T)]
    P = lol_amd.Plan(pps, qs)
    a = torch.stack([torch.randint(0, q, (B, P.n), dtype=torch.int64, device="cuda") for q in qs], dim=-1).contiguous()
    b = a.clone()
    for op in ops:
        f = (lambda: P.polymul(a, b, out=b)) if op=="polymul" else (lambda: getattr(P, op)(a))
        f(); torch.cuda.synchronize()
        t0=time.perf_counter(); it=5
        for _ in range(it): f()
        torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/it
        byt = (3 if op=="polymul" else 2)*B*P.n*T*8
        print(f"m={m} n={P.n} T={T} B={B} {op:8s} {dt*1e3:9.3f} ms  {B/dt/1e6:8.3f} M/s  {byt/dt/1e9:8.1f} GB/s ({byt/dt/8e10:.1f}% of 8TB/s)", flush=True)
run(15015, 2**60, 1024)
run(15015, 2**30, 1024)
run(64*27, 3000, 8192, ops=("crt","l","mulGPow"))
run(2048*7, 2**20, 1024, ops=("crt","l"))
