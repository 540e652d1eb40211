#!/usr/bin/env python3
"""tools/bench_pipelines.py — throughput of the SymmSHE pipeline kernels on one MI355X
(BASELINE configs 3 and 5 shapes).  Operands resident in HBM, HIP events on
the launch stream.  Prints one JSON object per line; `alg_bytes` is the compulsory traffic
of the *fused ideal* (each input slab read once, each output written once)."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lol_amd  # noqa: E402


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for s, e in ev:
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    return sum(s.elapsed_time(e) for s, e in ev) / iters


def rnd(gen, qs, *shape):
    return torch.stack([torch.randint(0, q, shape, dtype=torch.int64, device="cuda", generator=gen) for q in qs], dim=-1)


def report(name, cfg, ms, items, alg_bytes, note=None):
    d = {"op": name, "config": cfg, "ms": round(ms, 4), "items_per_s": round(items / ms * 1e3, 1),
         "alg_GBps": round(alg_bytes / ms / 1e6, 1), "frac_of_8TBps": round(alg_bytes / ms / 1e6 / 8000, 4)}
    if note:
        d["note"] = note
    print(json.dumps(d), flush=True)


def good_qs(m, lower, T):
    out, lo = [], lower
    for _ in range(T):
        q = lol_amd.good_q(m, lo)
        out.append(q); lo = q
    return out


def main():
    gen = torch.Generator(device="cuda"); gen.manual_seed(1)
    L = lol_amd.lib()
    st = torch.cuda.current_stream().cuda_stream
    # ---- config 3: ciphertext product, m = 2^15, T = 4, ~59-bit moduli ------------------
    qs = good_qs(2 ** 15, 2 ** 59, 4)
    P = lol_amd.Plan([(2, 15)], qs)
    B = 256
    ops = [rnd(gen, qs, B, P.n) for _ in range(4)]
    outs = [torch.empty_like(ops[0]) for _ in range(3)]
    slab = B * P.n * P.T * 8
    ptr = lambda t: t.data_ptr()
    ms = timeit(lambda: L.lolhip_ctmul_crt_batch(P._h, st, *map(ptr, ops + outs), B))
    report("ctmul_crt", f"m=2^15 T=4 59-bit B={B}", ms, B, 7 * slab)
    # the same product op by op (what a Tensor-method-at-a-time backend does): 4 mul, 1 add, 3 mulGCRT
    def unfused():
        a, b_, c_, d_ = (o.clone() for o in (ops[0], ops[0], ops[1], ops[1]))
        P.mul(a, ops[2]); P.mul(b_, ops[3]); P.mul(c_, ops[2]); P.mul(d_, ops[3])
        b_ += c_
        P.mulGCRT(a); P.mulGCRT(b_); P.mulGCRT(d_)
    ms_u = timeit(unfused, iters=5)
    report("ctmul_crt_op_by_op", f"m=2^15 T=4 59-bit B={B}", ms_u, B, 7 * slab)
    del ops, outs
    # ---- config 5: key switch, m' = 2048, q = (1017857, 1032193), TrivGad; and at n = 8192 -----
    # (pps, moduli, batch): config 5's ring; the same at n = 8192; the reference's non-2-power key-switch benchmark
    # F64*F9*F25 with Zq (1008001 ** 1065601) (lol-apps Benchmarks/Default.hs:49)
    for pps5, qs5, B in (([(2, 11)], [1017857, 1032193], 8192), ([(2, 14)], good_qs(2 ** 14, 2 ** 20, 2), 1024),
                         ([(2, 6), (3, 2), (5, 2)], [1008001, 1065601], 2048),
                         ([(2, 6), (3, 2), (5, 2)], [1008001, 1065601], 8192)):      # the same with a digit slab beyond the Infinity Cache
        P = lol_amd.Plan(pps5, qs5)
        for base in (0, 256):
            Ld = P.decomposeLen(base)
            c2 = rnd(gen, qs5, B, P.n)
            add = torch.stack([rnd(gen, qs5, B, P.n) for _ in range(2)])
            hint = rnd(gen, qs5, Ld, 2, P.n)
            work = torch.empty((Ld, B, P.n, P.T), dtype=torch.int64, device="cuda")
            out = torch.empty_like(add)
            slab = B * P.n * P.T * 8
            ms = timeit(lambda: L.lolhip_keyswitch_batch(P._h, st, ptr(c2), base, ptr(hint), 2, ptr(add), ptr(out), ptr(work), B))
            report("keyswitch", f"m={P.m} T=2 q~2^20 base={base} L={Ld} B={B}", ms, B, 5 * slab)
            L.lolhip_debug_set(b"KEYSWITCH_UNFUSED", 1)
            ms = timeit(lambda: L.lolhip_keyswitch_batch(P._h, st, ptr(c2), base, ptr(hint), 2, ptr(add), ptr(out), ptr(work), B))
            L.lolhip_debug_set(b"KEYSWITCH_UNFUSED", 0)
            report("keyswitch_three_launches", f"m={P.m} T=2 q~2^20 base={base} L={Ld} B={B}", ms, B, 5 * slab)
            ms = timeit(lambda: L.lolhip_decompose_batch(P._h, st, ptr(c2), base, ptr(work), B))
            report("  decompose", f"L={Ld}", ms, B, (1 + Ld) * slab)
            ms = timeit(lambda: L.lolhip_crt_batch(P._h, st, ptr(work), Ld * B))
            ic = "digit slab %d MiB: re-read from the 256 MiB Infinity Cache when it fits (not an HBM rate)" % (Ld * slab >> 20) if Ld * slab <= (256 << 20) else None
            report("  crt(digits)", f"L={Ld}", ms, B * Ld, 2 * Ld * slab, note=ic)
            ms = timeit(lambda: L.lolhip_knapsack_batch(P._h, st, ptr(work), Ld, ptr(hint), 2, ptr(add), ptr(out), B))
            report("  knapsack", f"L={Ld} K=2", ms, B, (Ld + 4) * slab)
    # ---- rescale: drop the first of four 59-bit moduli at m = 2^15 --------------------------
    qs = good_qs(2 ** 15, 2 ** 59, 4)
    P = lol_amd.Plan([(2, 15)], qs)
    B = 256
    c = rnd(gen, qs, B, P.n)
    out = torch.empty((B, P.n, 3), dtype=torch.int64, device="cuda")
    ms = timeit(lambda: L.lolhip_rescale_drop_batch(P._h, st, ptr(c), ptr(out), B))
    report("rescale_drop", f"m=2^15 T=4->3 B={B}", ms, B, B * P.n * 7 * 8)

    streaming(gen)


def streaming(gen):
    """the HBM-bound Tensor members of SURVEY 8(a) a6, a12-a15: mulRq, mulGCRT, twace*/embed* (config 5's ring pair)"""
    if "--no-streaming" in sys.argv:
        return
    q61 = lol_amd.good_q(2 ** 14, 2 ** 60)
    for cfg, pps, qs, B in (("m=2^14 T=1 61-bit", [(2, 14)], [q61], 4096), ("m=2^11 T=2 q~2^20", [(2, 11)], [1017857, 1032193], 32768)):
        P = lol_amd.Plan(pps, qs)
        a, b = rnd(gen, qs, B, P.n), rnd(gen, qs, B, P.n)
        slab = B * P.n * P.T * 8
        report("mulRq", f"{cfg} B={B}", timeit(lambda: P.mul(a, b)), B, 3 * slab)
        report("mulGCRT", f"{cfg} B={B}", timeit(lambda: P.mulGCRT(a)), B, 2 * slab)
    qs, B = [1017857, 1032193], 8192          # both are 1 mod 14336
    lo, hi = lol_amd.Plan([(2, 11)], qs), lol_amd.Plan([(2, 11), (7, 1)], qs)
    E = lol_amd.Ext(lo, hi)
    x_lo, x_hi = rnd(gen, qs, B, lo.n), rnd(gen, qs, B, hi.n)
    o_lo, o_hi = torch.empty_like(x_lo), torch.empty_like(x_hi)
    byts = B * (lo.n + hi.n) * 2 * 8
    # twacePowDec reads only phi(m) of the phi(m') coefficients (every 6th here, 16 bytes each: whole 64-byte sectors
    # are fetched): compulsory traffic = n coefficients in and out, NOT the (n + n') of the other members
    byts_twpd = B * 2 * lo.n * 2 * 8
    cfg = f"2048 -> 14336 T=2 B={B}"
    for name, fn, bb, note in (("embedPow", lambda: E.embedPow(x_lo, out=o_hi), byts, None), ("embedDec", lambda: E.embedDec(x_lo, out=o_hi), byts, None),
                               ("embedCRT", lambda: E.embedCRT(x_lo, out=o_hi), byts, None),
                               ("twacePowDec", lambda: E.twacePowDec(x_hi, out=o_lo), byts_twpd,
                                "2 n T 8 bytes per item; the gathered 16-byte pairs sit 96 bytes apart, so the sectors fetched are 4x the bytes used"),
                               ("twaceCRT", lambda: E.twaceCRT(x_hi, out=o_lo), byts, None)):
        report(name, cfg, timeit(fn), B, bb, note=note)


if __name__ == "__main__":
    main()
