#!/usr/bin/env python3
"""bench.py — cyclotomic poly-muls/sec (CRT . pointwise . CRT^-1), batched, on MI355X.

Workload (BASELINE.json configs[1]): m = 2^14 (n = 8192), q = first prime = 1 mod 2^14
above 2^60 (goodQs rule, ZqBasic.hs:71-73), batch = 4096 polynomials per GPU,
operands and results in the powerful basis, resident in HBM.  One step = one fused
poly-mul launch over the whole batch (c = crtInv(crt a * crt b)).

  python bench.py --gpus N --steps K --warmup W
N > 1 runs one rank per GPU.  Either the caller starts the ranks (torch.distributed.run sets
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*; WORLD_SIZE must then equal --gpus), or — plain
`python bench.py --gpus N` with no WORLD_SIZE in the environment — this process starts N fresh
child ranks itself BEFORE anything touches the GPU (launch_ranks below: the parent never
initialises HIP and never re-execs), relays rank 0's JSON line and exits non-zero if any rank fails.
The batch dimension shards with no data-path collective (every polynomial is independent),
so scaling is weak: each rank multiplies its own 4096-polynomial shard.

Prints ONE JSON line (rank 0).  `roofline.achieved` is algorithmic bytes
(3 * n * T * 8 per poly-mul) over the mean launch duration measured with
HIP events on the launch stream inside the timed region (one pair around the K launches).

Before the W warm-up steps the same step runs untimed for ~0.4 s, so that a short timed
region (the driver's --steps 20 is 12 ms) sees the clocks the chip holds under this load and
not its ramp.  After the timed region, N = 1 only and outside the metric, `secondary` carries
the rest of SURVEY.md 8(d)'s table — each leg {ms, items_per_s, alg_GBps, frac of 8 TB/s}: the
same launch in the other arithmetic classes, config 3's ciphertext product, config 4 at both
moduli, config 5's key switch and ring embedding, and the drop-in (host-pointer) symbols at
config 1 beside lol-cpp on one host core.  For N > 1 a `gather` object times the one optional
collective (all-gather of the result shards) and gives poly-muls/s with it included.
"""
from __future__ import annotations

import argparse
import json
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

M_INDEX = 1 << 14
N_COEF = M_INDEX // 2
BATCH = 4096
Q_LOWER = 1 << 60
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec


# ---------------------------------------------------------------------------------
# CPU baseline: the reference's own lol-cpp (oracle/_ref/libctensor.so), one PROCESS
# per core because its modulus is a process-global (types.h:59).
# ---------------------------------------------------------------------------------
def _cpu_worker(args):
    q, count, seed, kind = args
    import numpy as np
    from oracle import lolmath as lm
    from oracle.oracle import CTRef, CpuRef, Params
    P = Params([(2, 14)], [q])
    rng = np.random.default_rng(seed)
    a, b = P.random(rng, 1), P.random(rng, 1)
    eng = CTRef() if kind == "reference" else CpuRef()
    eng.polymul(P, a, b)  # warm
    t0 = time.perf_counter()
    for _ in range(count):
        eng.polymul(P, a, b)
    return time.perf_counter() - t0


def _usable_cores() -> int:
    """Cores this job may really use: affinity, capped by the cgroup CPU quota (the GPU box
    exposes 64 CPUs but grants a 1-GPU job a 16-core share)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 16))


def cpu_baseline(q: int, budget_s: float = 12.0):
    from oracle.oracle import CTREF_SO
    kind = "reference" if os.path.exists(CTREF_SO) else "port"
    cores = _usable_cores()
    # calibrate one poly-mul on one core, then size the sample to the budget
    t1 = _cpu_worker((q, 3, 1, kind)) / 3
    count = max(4, int(budget_s / max(t1, 1e-4)))
    ctx = mp.get_context("spawn")
    t0 = time.perf_counter()
    with ctx.Pool(cores) as pool:
        pool.map(_cpu_worker, [(q, count, 100 + i, kind) for i in range(cores)])
    wall = time.perf_counter() - t0
    # the pool wall time includes process start-up; use it (conservative for the GPU ratio)
    # the same workload at a modulus where lol-cpp is itself CORRECT (q < 2^31.5): same n, same code
    q30 = 1073872897
    t30 = _cpu_worker((q30, 3, 2, kind)) / 3
    return {
        "value": round(cores * count / wall, 2),
        "unit": "poly-muls/s",
        "cores": cores,
        "kind": kind,
        "single_core_ms_per_polymul": round(t1 * 1e3, 3),
        "single_core_ms_per_polymul_q30": round(t30 * 1e3, 3),
        "sample": (f"{cores} processes x {count} poly-muls (2 crt + mulRq + crtInv through lol-cpp's C ABI), "
                   f"n=8192, q={q}; timing only at this modulus: lol-cpp's Zq overflows for q > ~2^31.5 "
                   "(types.h:79-84), same instruction stream"),
    }


def _leg(ms, items, alg_bytes, **extra):
    d = {"ms": round(ms, 4), "items_per_s": round(items / ms * 1e3, 1), "alg_GBps": round(alg_bytes / ms / 1e6, 1),
         "frac": round(alg_bytes / ms / 1e6 / HBM_PEAK_GBS, 4)}
    d.update(extra)
    return d


def _check_rc(rc):
    if rc != 0:
        raise SystemExit(f"bench: liblolhip returned status {rc}")


def _good_qs(lol_amd, m, lower, T):
    out, lo = [], lower
    for _ in range(T):
        lo = lol_amd.good_q(m, lo)
        out.append(lo)
    return out


def secondary(lol_amd, torch, plan, a, c, B, n, T, stream):
    """Not the metric: the rest of SURVEY.md 8(d)'s table (N = 1 only).  ms per launch from HIP
    events on the launch stream; alg_GBps = compulsory bytes of the fused ideal / time."""
    st = stream.cuda_stream

    def timed(fn, iters=10, warm=2):
        for _ in range(warm):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(iters):
            fn()
        e1.record(stream)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters

    gen = torch.Generator(device="cuda")
    gen.manual_seed(7)

    def rnd(qs, *shape):
        return torch.stack([torch.randint(0, q, shape, dtype=torch.int64, device="cuda", generator=gen) for q in qs], dim=-1)

    slab = B * n * T * 8
    out = {}
    # ---- config 2's launch shape in the other arithmetic classes -----------------------------
    # the yardstick beside the 8 TB/s datasheet peak: liblolhip's own read-once/write-once copy of the same slab,
    # 16 bytes per lane (lolhip_copy_slab); torch's copy_ beside it.  Every HBM-bound leg below also carries
    # `vs_copy` = its algorithmic GB/s over this copy's.  (An IN-PLACE transform re-writes the lines it has just
    # read and can beat an out-of-place copy: crt_27bit does.)
    Lc = lol_amd.lib()
    out["hbm_copy"] = _leg(timed(lambda: _check_rc(Lc.lolhip_copy_slab(st, c.data_ptr(), a.data_ptr(), slab, 0))), B, 2 * slab,
                           workload="lolhip_copy_slab: 16 B per lane, one 16 KiB tile per workgroup, 256 MiB slab (read + write), same stream")
    out["hbm_copy_torch"] = _leg(timed(lambda: c.copy_(a)), B, 2 * slab, workload="torch copy_ of the same slab")
    c.copy_(a)
    out["crt_61bit"] = _leg(timed(lambda: plan.crt(c, stream=st)), B, 2 * slab)
    y = torch.empty_like(a)
    # 27bit: arithmetic class 4 (every q < 2^27, the reference's own benchmark moduli, Benchmarks/Default.hs:42-50);
    # 30bit: class 2; 31bit: class 3 at config 2's CT-valid modulus
    for name, q_ in (("27bit", lol_amd.good_q(M_INDEX, 1 << 26)), ("30bit", lol_amd.good_q(M_INDEX, 1 << 29)), ("31bit", 1073872897)):
        pq = lol_amd.Plan([(2, 14)], [q_])
        x, x2 = a % q_, (a + 1) % q_
        out[f"polymul_{name}"] = _leg(timed(lambda: pq.polymul(x, x2, out=y, stream=st)), B, 3 * slab, q=q_)
        y.copy_(x)
        out[f"crt_{name}"] = _leg(timed(lambda: pq.crt(y, stream=st)), B, 2 * slab, q=q_)
        del pq, x, x2
    del y
    L = lol_amd.lib()
    ptr = lambda t_: t_.data_ptr()
    # ---- config 3: m = 2^15, four ~59-bit moduli, ciphertext x ciphertext (SymmSHE.hs:444-449) ----
    qs3 = _good_qs(lol_amd, 1 << 15, 1 << 59, 4)
    P3 = lol_amd.Plan([(2, 15)], qs3)
    B3 = 256
    ops = [rnd(qs3, B3, P3.n) for _ in range(4)]
    outs = [torch.empty_like(ops[0]) for _ in range(3)]
    slab3 = B3 * P3.n * P3.T * 8
    ms = timed(lambda: L.lolhip_ctmul_crt_batch(P3._h, st, *map(ptr, ops + outs), B3))
    out["c3_ctmul_crt"] = _leg(ms, B3, 7 * slab3, workload="m=2^15 T=4 59-bit B=256: (g c0 d0, g(c0 d1 + c1 d0), g c1 d1), CRT basis in and out")

    def ring_product():           # powerful basis in and out: 4 crt, the fused product, 3 crtInv
        for o in ops:
            P3.crt(o, stream=st)
        L.lolhip_ctmul_crt_batch(P3._h, st, *map(ptr, ops + outs), B3)
        for o in outs:
            P3.crtInv(o, stream=st)
    ms = timed(ring_product, iters=5, warm=1)
    out["c3_ctmul_pow"] = _leg(ms, B3, 7 * slab3, workload="the same from and to the powerful basis: 4 crt + product + 3 crtInv (8 launches)")
    # config 3's ring and moduli through the key switch (SymmSHE.hs:361-371): 64-bit moduli take the three-launch path
    # (decompose -> L*B crt -> knapsack); B = 64 ciphertexts keeps the digit slab at 1 GiB
    B3k = 64
    c2k = rnd(qs3, B3k, P3.n)
    addk = torch.stack([rnd(qs3, B3k, P3.n) for _ in range(2)])
    resk = torch.empty_like(addk)
    Ldk = P3.decomposeLen(0)
    hintk = rnd(qs3, Ldk, 2, P3.n)
    workk = torch.empty((Ldk, B3k, P3.n, P3.T), dtype=torch.int64, device="cuda")
    ms = timed(lambda: _check_rc(L.lolhip_keyswitch_batch(P3._h, st, ptr(c2k), 0, ptr(hintk), 2, ptr(addk), ptr(resk), ptr(workk), B3k)), iters=5, warm=1)
    out["c3_keyswitch_trivgad"] = _leg(ms, B3k, 5 * B3k * P3.n * P3.T * 8, digits=Ldk,
                                       workload="keySwitchQuadCirc body at m=2^15 T=4 59-bit, TrivGad, B=64: three launches (no fused kernel for 64-bit moduli)")
    del ops, outs, P3, c2k, addk, resk, hintk, workk
    # ---- config 4: m = 15015, batch 1024, q just above 2^30 and just above 2^60 ---------------
    pps4 = lol_amd.factor_pps(15015)
    for name, lower in (("q30", 1 << 30), ("q60", 1 << 60)):
        q4 = lol_amd.good_q(15015, lower)
        P4 = lol_amd.Plan(pps4, [q4])
        B4 = 1024
        x, x2 = rnd([q4], B4, P4.n), rnd([q4], B4, P4.n)
        y = torch.empty_like(x)
        slab4 = B4 * P4.n * 8
        out[f"c4_polymul_{name}"] = _leg(timed(lambda: P4.polymul(x, x2, out=y, stream=st)), B4, 3 * slab4, q=q4)
        out[f"c4_crt_{name}"] = _leg(timed(lambda: P4.crt(x, stream=st)), B4, 2 * slab4, q=q4)
        del P4, x, x2, y
    # ---- the reference's own benchmark index shapes, m = 2^e * odd (lol/.../Benchmarks/Default.hs:42-50:
    # 64*9*25; lol-apps tunnelling chain 128*7*13), a reference-sized modulus, batch 8192 ----
    for mref in (14400, 11648):
        qr = lol_amd.good_q(mref, 1 << 26)
        Pr = lol_amd.Plan(lol_amd.factor_pps(mref), [qr])
        Br = 8192
        x, x2 = rnd([qr], Br, Pr.n), rnd([qr], Br, Pr.n)
        y = torch.empty_like(x)
        slabr = Br * Pr.n * 8
        out[f"ref_m{mref}_polymul"] = _leg(timed(lambda: Pr.polymul(x, x2, out=y, stream=st)), Br, 3 * slabr, q=qr, n=Pr.n)
        out[f"ref_m{mref}_crt"] = _leg(timed(lambda: Pr.crt(x, stream=st)), Br, 2 * slabr, q=qr, n=Pr.n)
        del Pr, x, x2, y
    # ---- config 5: key switch at m' = 2048, q = (1017857, 1032193); ring embedding 2048 -> 14336 ----
    qs5 = [1017857, 1032193]
    P5 = lol_amd.Plan([(2, 11)], qs5)
    B5 = 8192                                  # one GPU's shard of the 65536-ciphertext batch
    slab5 = B5 * P5.n * P5.T * 8
    c2 = rnd(qs5, B5, P5.n)
    add = torch.stack([rnd(qs5, B5, P5.n) for _ in range(2)])
    res = torch.empty_like(add)
    for name, base in (("trivgad", 0), ("base256", 256)):
        Ld = P5.decomposeLen(base)
        hint = rnd(qs5, Ld, 2, P5.n)
        work = torch.empty((Ld, B5, P5.n, P5.T), dtype=torch.int64, device="cuda")
        ms = timed(lambda: L.lolhip_keyswitch_batch(P5._h, st, ptr(c2), base, ptr(hint), 2, ptr(add), ptr(res), ptr(work), B5))
        out[f"c5_keyswitch_{name}"] = _leg(ms, B5, 5 * slab5, digits=Ld, workload="keySwitchQuadCirc body: c2 in, 2 addends in, 2 out (SymmSHE.hs:361-371), m'=2048 T=2 B=8192")
        del hint, work
    del c2, add, res
    P5h = lol_amd.Plan(lol_amd.factor_pps(2048 * 7), qs5)
    X = lol_amd.Ext(P5, P5h)
    lo = rnd(qs5, B5, P5.n)
    hi = torch.empty((B5, P5h.n, 2), dtype=torch.int64, device="cuda")
    ms = timed(lambda: X.embedCRT(lo, out=hi, stream=st))
    out["c5_embed_crt"] = _leg(ms, B5, (P5.n + P5h.n) * 2 * 8 * B5, workload="embedCRT 2048 -> 14336 (Extension.hs:81-85), T=2 B=8192")
    del X, lo, hi, P5h, P5
    copy_gbps = out["hbm_copy"]["alg_GBps"]
    for k, v in out.items():
        if isinstance(v, dict) and "alg_GBps" in v and k not in ("hbm_copy",):
            v["vs_copy"] = round(v["alg_GBps"] / copy_gbps, 3)
    return out


def dropin_c1():
    """Config 1 through the drop-in symbols (host pointers, one polynomial per call, the path the
    unchanged CT shim takes: Backend.hs:304-337) beside lol-cpp itself on one host core."""
    import ctypes as C

    import numpy as np

    import lol_amd
    from oracle import lolmath as lm
    from oracle.oracle import CTREF_SO, CpuRef, Params

    m, q = 1024, 12289
    P = Params([(2, 10)], [q])

    class PP(C.Structure):
        _fields_ = [("prime", C.c_int16), ("exponent", C.c_int16)]
    pe = (PP * 1)()
    pe[0].prime, pe[0].exponent = 2, 10
    qa = np.array([q], dtype=np.int64)
    ru = [np.array(t, dtype=np.int64) for t in P.ru]
    rui = [np.array(t, dtype=np.int64) for t in P.ruinv]
    rup = (C.c_void_p * 1)(ru[0].ctypes.data)
    ruip = (C.c_void_p * 1)(rui[0].ctypes.data)
    mh = np.array(P.mhatinv, dtype=np.int64)
    rng = np.random.default_rng(1)
    a0, b0 = P.random(rng, 1)[0], P.random(rng, 1)[0]
    want = CpuRef().polymul(P, a0[None], b0[None]).reshape(P.n, 1)

    def per_call(fn, reps):
        fn()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        return (time.perf_counter() - t0) / reps * 1e6

    def measure(path, totm_type):
        """The SAME ctypes calls against either library: the symbols are drop-in identical
        (lol-cpp declares totm as int32 hDim_t, types.h:22; Haskell and liblolhip pass int64)."""
        Lh = C.CDLL(path)
        i16, vp = C.c_int16, C.c_void_p
        Lh.tensorCRTRq.argtypes = [i16, vp, totm_type, vp, i16, vp, vp]
        Lh.tensorCRTInvRq.argtypes = [i16, vp, totm_type, vp, i16, vp, vp, vp]
        Lh.mulRq.argtypes = [i16, vp, vp, totm_type, vp]
        for nm in ("tensorCRTRq", "tensorCRTInvRq", "mulRq"):
            getattr(Lh, nm).restype = None

        def crt(y):
            Lh.tensorCRTRq(1, y.ctypes.data, P.n, pe, 1, rup, qa.ctypes.data)

        def polymul():
            a, b = a0.copy(), b0.copy()
            crt(a); crt(b)
            Lh.mulRq(1, a.ctypes.data, b.ctypes.data, P.n, qa.ctypes.data)
            Lh.tensorCRTInvRq(1, a.ctypes.data, P.n, pe, 1, ruip, mh.ctypes.data, qa.ctypes.data)
            return a
        ok = bool(np.array_equal(polymul(), want))
        y = a0.copy()
        return round(per_call(lambda: crt(y), 300), 1), round(per_call(polymul, 100), 1), ok

    g_crt, g_pm, g_ok = measure(lol_amd.lib_path(), C.c_int64)
    res = {"workload": "m=1024 n=512 q=12289, one polynomial per call through the drop-in symbols (host pointers; BASELINE.json configs[0])",
           "gpu_us_per_tensorCRTRq": g_crt, "gpu_us_per_polymul": g_pm, "parity_ok": g_ok}
    if os.path.exists(CTREF_SO):
        c_crt, c_pm, c_ok = measure(CTREF_SO, C.c_int32)
        res.update({"lolcpp_us_per_tensorCRTRq_one_core": c_crt, "lolcpp_us_per_polymul_one_core": c_pm, "lolcpp_parity_ok": c_ok})
    return res


def timed_steps(step, steps, dist, sync, device, before=None, after=None):
    """The contract's timed region: barrier + device sync on both sides of EXACTLY `steps` steps,
    MAX over ranks of the local wall time.  `dist` is torch.distributed or None (one rank)."""
    import torch
    if dist is not None:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    if before:
        before()
    for _ in range(steps):
        step()
    if after:
        after()
    sync()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    return elapsed


def launcher_selftest(args, rank, world):
    """CPU-only: the rank start-up, rendezvous, barrier/MAX timing and rank-0 reporting of the N > 1
    path on gloo, with a step that launches nothing.  tests/test_dist.py runs it at N = 2."""
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from lol_amd.dist import shard_range
    lo, hi = shard_range(args.batch * world, rank, world)       # weak scaling: every rank owns args.batch items
    acc = [0]

    def step():
        acc[0] += hi - lo
    for _ in range(args.warmup):
        step()
    acc[0] = 0
    elapsed = timed_steps(step, args.steps, dist, lambda: None, "cpu")
    total = torch.tensor([acc[0]], dtype=torch.int64)
    if dist is not None:
        dist.all_reduce(total)
    if rank == 0:
        print(json.dumps({"metric": "launcher-selftest (no kernel launched; NOT a measurement)", "value": None,
                          "n_gpus": world, "world_size": dist.get_world_size() if dist else 1,
                          "backend": dist.get_backend() if dist else None, "steps": args.steps, "warmup": args.warmup,
                          "items_all_ranks": int(total.item()), "elapsed_s": round(elapsed, 6)}), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def launch_ranks(argv, n, timeout_s=None):
    """Start n child ranks of this script (fresh interpreters; RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* in their environment), wait for all of them, relay rank 0's stdout.  Runs before any
    GPU call: the parent only waits.  Returns the exit code for the whole job."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs on this driver
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    t_end = None if timeout_s is None else time.time() + timeout_s
    out0, codes = b"", [None] * n
    try:
        out0 = procs[0].communicate(timeout=timeout_s)[0]
        codes[0] = procs[0].returncode
        for r in range(1, n):
            left = None if t_end is None else max(1.0, t_end - time.time())
            codes[r] = procs[r].wait(timeout=left)
    except subprocess.TimeoutExpired:
        pass
    finally:
        for pr in procs:            # exactly the children started above, by PID
            if pr.poll() is None:
                pr.kill()
                pr.wait()
    sys.stdout.write(out0.decode("utf-8", "replace"))
    sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad:
        print(f"bench: ranks failed or timed out (rank, exit code): {bad}", file=sys.stderr)
        return 1
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--clock-warmup-s", type=float, default=0.4)
    ap.add_argument("--launcher-selftest", action="store_true",
                    help="CPU-only check of the rank launcher and the rendezvous/timing protocol (gloo, no kernel): "
                         "prints a line whose metric says so; never a measurement")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(sys.argv[1:], args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench: --gpus {args.gpus} but WORLD_SIZE={world}: start one rank per GPU "
                         "(or leave WORLD_SIZE unset and let bench.py start them)")
    if args.launcher_selftest:
        return launcher_selftest(args, rank, world)

    import numpy as np
    import torch

    import lol_amd

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (liblolhip has no CPU fallback)")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"bench: rank {rank} wants cuda:{local_rank} but the node exposes {torch.cuda.device_count()} GPU(s)")
    torch.cuda.set_device(local_rank)
    dist = None
    backend = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        backend = dist.get_backend()
        if dist.get_world_size() != args.gpus:
            raise SystemExit("bench: process group size differs from --gpus")
    n_gpus = world

    q = lol_amd.good_q(M_INDEX, Q_LOWER)
    plan = lol_amd.Plan([(2, 14)], [q])
    B, n, T = args.batch, plan.n, plan.T
    gen = torch.Generator(device="cuda")
    gen.manual_seed(2 + rank)
    # uniform residues in [0, q): synthetic operands already resident in HBM
    a = torch.randint(0, q, (B, n, T), dtype=torch.int64, device="cuda", generator=gen)
    b = torch.randint(0, q, (B, n, T), dtype=torch.int64, device="cuda", generator=gen)
    c = torch.empty_like(a)
    stream = torch.cuda.current_stream()

    def step():
        plan.polymul(a, b, out=c, stream=stream.cuda_stream)

    # clock warm-up (untimed, not counted in --warmup): the chip's clocks settle under THIS load
    t_w = time.perf_counter()
    while time.perf_counter() - t_w < args.clock_warmup_s:
        for _ in range(20):
            step()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    # one HIP-event pair on the launch stream around the K launches (they queue back to back;
    # an event pair per launch would put a marker packet between every two kernels)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    elapsed = timed_steps(step, args.steps, dist, torch.cuda.synchronize, "cuda",
                          before=lambda: ev0.record(stream), after=lambda: ev1.record(stream))
    kern_ms = ev0.elapsed_time(ev1) / max(1, args.steps)

    # ---- N > 1: the one optional collective, all-gather of the result shards ----
    gather = None
    if dist is not None:
        from lol_amd.dist import allgather_batch
        full = allgather_batch(c, B * world)             # warm (RCCL channel setup)
        torch.cuda.synchronize()
        dist.barrier()
        tg0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            full = allgather_batch(c, B * world)
        torch.cuda.synchronize()
        dist.barrier()
        tg = torch.tensor([(time.perf_counter() - tg0) / reps], dtype=torch.float64, device="cuda")
        dist.all_reduce(tg, op=dist.ReduceOp.MAX)
        gather = float(tg.item())
        del full

    # ---- correctness of what was timed: strided sample against the CPU oracle -----
    parity = None
    if rank == 0:
        from oracle.oracle import CpuRef, Params
        ref = Params([(2, 14)], [q])
        # first and last polynomial of every dispatch round (1024 workgroups resident at a time), both sides
        idx = sorted({i for i in (0, 1023, 1024, 2047, 2048, 3071, 3072, B - 1, B // 2) if 0 <= i < B})
        want = CpuRef().polymul(ref, a[idx].cpu().numpy(), b[idx].cpu().numpy())
        parity = bool(np.array_equal(c[idx].cpu().numpy(), want.reshape(len(idx), n, T)))
        if not parity:
            raise SystemExit("bench: GPU poly-mul differs from the oracle — result invalid")

    if rank == 0:
        value = n_gpus * B * args.steps / elapsed
        alg_bytes = 3 * n * T * 8 * B                      # per launch
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9     # GB/s
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tfile):
            try:
                traffic = json.load(open(tfile)).get("k_pow2_polymul_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "cyclotomic poly-muls/sec (CRT·pointwise·CRT⁻¹), batched; bit-exact vs CT",
            "value": round(value, 1),
            "unit": "poly-muls/s",
            "n_gpus": n_gpus,
            "world_size": dist.get_world_size() if dist is not None else 1,
            "backend": (backend + " (RCCL)") if backend == "nccl" else backend,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {"workload": "m=2^14 (n=8192), 61-bit prime q, batch=4096 poly-muls per GPU (BASELINE.json configs[1])",
                       "m": M_INDEX, "n": n, "q": q, "tupSize": T, "batch_per_gpu": B,
                       "sharding": f"batch x{n_gpus}, no data-path collective"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "traffic_source": "stored: profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of this kernel, separate passes); not measured by this run",
                         "kernel": "k_pow2<13,2,1,true> (fused crt,crt,mul,crtInv; 16 B per lane)", "kernel_ms": round(kern_ms, 4),
                         "algorithmic_bytes_per_launch": alg_bytes},
            "parity_sample_ok": parity,
            "parity_note": ("bit-exact vs the pinned CPU restatement (oracle/cpu_ref.c, itself pinned on lol-cpp's golden vectors); "
                            "lol-cpp's own Zq overflows at this 61-bit modulus (types.h:79-84) - the figure at a modulus "
                            "where CT is literally checkable is secondary.polymul_31bit"),
        }
        if gather is not None:
            shard_bytes = B * n * T * 8
            out["gather"] = {"ms": round(gather * 1e3, 4), "shard_bytes": shard_bytes,
                             "GBps_per_rank_received": round((world - 1) * shard_bytes / gather / 1e9, 1),
                             "poly_muls_per_s_with_gather": round(n_gpus * B / (elapsed / args.steps + gather), 1),
                             "note": "all-gather of every rank's result shard after one step (lol_amd.dist.allgather_batch, RCCL); optional: results consumed where produced need no collective"}
        if n_gpus == 1 and not args.no_secondary:
            out["secondary"] = secondary(lol_amd, torch, plan, a, c, B, n, T, stream)
            out["secondary"]["dropin_c1"] = dropin_c1()
        if n_gpus == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(q)
        print(json.dumps(out, ensure_ascii=False), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
