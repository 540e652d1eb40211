#!/usr/bin/env python3
"""bench.py — cyclotomic poly-muls/sec (CRT . pointwise . CRT^-1), batched, on MI355X.

Workload (BASELINE.json configs[1]): m = 2^14 (n = 8192), q = first prime = 1 mod 2^14
above 2^60 (goodQs rule, ZqBasic.hs:71-73), batch = 4096 polynomials per GPU,
operands and results in the powerful basis, resident in HBM.  One step = one fused
poly-mul launch over the whole batch (c = crtInv(crt a * crt b)).

  python bench.py --gpus N --steps K --warmup W
N > 1 is launched by the driver through torch.distributed.run (one rank per GPU); the
batch dimension shards with no data-path collective (every polynomial is independent),
so scaling is weak: each rank multiplies its own 4096-polynomial shard.

Prints ONE JSON line (rank 0).  `roofline.achieved` is algorithmic bytes
(3 * n * T * 8 per poly-mul) over the mean launch duration measured with
HIP events on the launch stream inside the timed region (one pair around the K launches).
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


def secondary(lol_amd, torch, plan, a, c, B, n, T, stream):
    """Not the metric: the same launch shape in the other regimes, for context (N = 1 only).
    ms per launch from HIP events; GB/s = algorithmic bytes / time."""
    def timed(fn, iters=10):
        fn()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
        for s_, e_ in ev:
            s_.record(stream); fn(); e_.record(stream)
        torch.cuda.synchronize()
        return sum(s_.elapsed_time(e_) for s_, e_ in ev) / iters
    slab = B * n * T * 8
    out = {}
    c.copy_(a)
    ms = timed(lambda: plan.crt(c, stream=stream.cuda_stream))
    out["crt_61bit"] = {"ms": round(ms, 4), "GBps": round(2 * slab / ms / 1e6, 1)}
    q30 = lol_amd.good_q(M_INDEX, 1 << 29)
    p30 = lol_amd.Plan([(2, 14)], [q30])
    x = a % q30
    y = torch.empty_like(x)
    x2 = (a + 1) % q30
    ms = timed(lambda: p30.polymul(x, x2, out=y, stream=stream.cuda_stream))
    out["polymul_30bit"] = {"ms": round(ms, 4), "poly_muls_per_s": round(B / ms * 1e3, 1), "GBps": round(3 * slab / ms / 1e6, 1), "q": q30}
    y.copy_(x)
    ms = timed(lambda: p30.crt(y, stream=stream.cuda_stream))
    out["crt_30bit"] = {"ms": round(ms, 4), "GBps": round(2 * slab / ms / 1e6, 1)}
    q31 = 1073872897            # config 2's CT-valid modulus: just above 2^30
    p31 = lol_amd.Plan([(2, 14)], [q31])
    x, x2 = a % q31, (a + 1) % q31
    ms = timed(lambda: p31.polymul(x, x2, out=y, stream=stream.cuda_stream))
    out["polymul_31bit"] = {"ms": round(ms, 4), "poly_muls_per_s": round(B / ms * 1e3, 1), "GBps": round(3 * slab / ms / 1e6, 1), "q": q31}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch

    import lol_amd

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (liblolhip has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
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

    for _ in range(args.warmup):
        step()
    # one HIP-event pair on the launch stream around the K launches (they queue back to back;
    # an event pair per launch would put a marker packet between every two kernels)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record(stream)
    for _ in range(args.steps):
        step()
    ev1.record(stream)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    kern_ms = ev0.elapsed_time(ev1) / max(1, args.steps)

    # ---- correctness of what was timed: strided sample against the CPU oracle -----
    parity = None
    if rank == 0:
        from oracle.oracle import CpuRef, Params
        ref = Params([(2, 14)], [q])
        idx = list(range(0, B, max(1, B // 4)))[:4]
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
                         "kernel": "k_pow2<13,2> (fused crt,crt,mul,crtInv)", "kernel_ms": round(kern_ms, 4),
                         "algorithmic_bytes_per_launch": alg_bytes},
            "parity_sample_ok": parity,
        }
        if n_gpus == 1:
            out["secondary"] = secondary(lol_amd, torch, plan, a, c, B, n, T, stream)
        if n_gpus == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(q)
        print(json.dumps(out, ensure_ascii=False), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
