#!/usr/bin/env python3
"""tools/collect_profiles.py — condense gpurun_out/refresh/ (written by tools/refresh_profiles.sh on
the GPU box) into the committed files under profiles/.  Delete the local gpurun_out/refresh/ before
the gpurun call: results are merged into it, and stale files from an earlier run would be averaged in."""
import csv, glob, json, os, re, sys, collections

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "refresh")
DST = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"


def pmc(dirname, pat="k_pow2"):
    agg = collections.defaultdict(list)
    for f in glob.glob(os.path.join(SRC, dirname, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}


# 1. bench line + kernel stats of the same command
bench = json.loads(open(os.path.join(SRC, "bench.json")).read().strip().splitlines()[-1])
rows = []
for f in glob.glob(os.path.join(SRC, "prof", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append([re.sub(r"\(.*", "", r["Name"])[:100], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
rows.sort(key=lambda r: -float(r[2]))
with open(os.path.join(DST, f"{tag}_bench_kernel_stats.csv"), "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["kernel", "calls", "total_ns", "avg_ns", "pct", "min_ns", "max_ns"])
    w.writerows(rows)

# 2. HBM traffic, calibrated on crt (known read volume), separate passes per counter
B, n = 4096, 8192
known_kb = B * n * 8 / 1024
f_crt, w_crt = pmc("pmc_crt_FETCH_SIZE")["FETCH_SIZE"], pmc("pmc_crt_WRITE_SIZE")["WRITE_SIZE"]
f_pm, w_pm = pmc("pmc_polymul_FETCH_SIZE")["FETCH_SIZE"], pmc("pmc_polymul_WRITE_SIZE")["WRITE_SIZE"]
factor = known_kb / f_crt[0]
read_b, write_b = f_pm[0] * factor * 1024, w_pm[0] * 1024
alg = 3 * n * 8 * B
traffic = {
    "guide_x2_hbm_bytes_per_launch": int(f_pm[0] * 2 * 1024 + write_b),     # MI355X_MICROARCH.md: FETCH_SIZE x 2 for wide streaming reads
    "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (MI355X_MICROARCH.md HBM section); values are KB. "
              "FETCH_SIZE under-reports wide streaming reads on gfx950; the factor is CALIBRATED on k_pow2<13,0> (crt in place), whose read "
              "volume is known exactly (4096*8192*8 B). WRITE_SIZE needs no correction (crt writes exactly that volume and WRITE_SIZE reports it).",
    "command": "rocprofv3 --pmc FETCH_SIZE --output-format csv -- tools/bench_kernels 14 1 4096 {crt,polymul} 5  (and --pmc WRITE_SIZE); tools/refresh_profiles.sh",
    "calibration_kernel": {"name": "k_pow2<13,0,1>", "known_read_KB": known_kb, "FETCH_SIZE_KB": f_crt[0], "factor": factor, "WRITE_SIZE_KB": w_crt[0]},
    "k_pow2_polymul": {"FETCH_SIZE_KB": f_pm[0], "WRITE_SIZE_KB": w_pm[0], "dispatches_averaged": f_pm[1], "read_bytes": int(read_b), "write_bytes": int(write_b),
                       "note": "round 2: the 64-bit fused kernel no longer spills (120 VGPRs); round 1 wrote one VGPR pair per thread to scratch (1.04x)"},
    "k_pow2_polymul_hbm_bytes_per_launch": int(read_b + write_b),
    "algorithmic_bytes_per_launch": alg,
    "traffic_over_algorithmic": round((read_b + write_b) / alg, 4),
}
json.dump(traffic, open(os.path.join(DST, "pmc_traffic.json"), "w"), indent=1)
bench["roofline"]["traffic"] = traffic["k_pow2_polymul_hbm_bytes_per_launch"]
json.dump(bench, open(os.path.join(DST, f"{tag}_bench.json"), "w"), ensure_ascii=False)

# 3. SQ counters of the fused kernel
sq = {}
for d in ("pmc_sq1", "pmc_sq2"):
    sq.update({k: v[0] for k, v in pmc(d).items()})
with open(os.path.join(DST, f"{tag}_pmc_polymul_sq.txt"), "w") as fh:
    fh.write("# rocprofv3 --pmc (two passes), tools/bench_kernels 14 1 4096 polymul 5; mean per dispatch of k_pow2<13,2,1>\n")
    for k in sorted(sq):
        fh.write(f"{k:28s} {sq[k]:.6g}\n")
    if "SQ_LDS_BANK_CONFLICT" in sq and "SQ_LDS_IDX_ACTIVE" in sq:
        fh.write(f"# LDS bank-conflict cycles / LDS active cycles = {sq['SQ_LDS_BANK_CONFLICT'] / sq['SQ_LDS_IDX_ACTIVE']:.3f}\n")
    if "SQ_ACTIVE_INST_VALU" in sq and "SQ_WAVE_CYCLES" in sq:
        fh.write(f"# VALU-active share of wave-cycles = {sq['SQ_ACTIVE_INST_VALU'] / sq['SQ_WAVE_CYCLES']:.3f} (4 waves/SIMD: 0.25 = VALU always busy)\n")

# 3b. config 4 (mixed radix): kernel trace, HBM traffic, SQ counters per (op, modulus)
c4 = {"workload": "m = 15015 (n = 5760), batch 1024, tools/bench_kernels m15015 1 1024 <op> <iters> <qbits>",
      "traffic_note": "FETCH_SIZE x 2 (MI355X_MICROARCH.md HBM section; 8-byte-per-lane buffer loads, so also quoted raw), WRITE_SIZE as read; KB per launch"}
for qb in (30, 60):
    for op in ("crt", "polymul"):
        ent = {}
        for f in glob.glob(os.path.join(SRC, f"c4_kt_{op}_{qb}", "**", "*kernel_stats.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if "k_mixed" in r["Name"]:
                    ent.update({"kernel": (re.search(r"k_mixed<[^>]*>", r["Name"]) or [r["Name"][:60]])[0], "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]),
                                "min_ns": float(r["MinNs"]), "max_ns": float(r["MaxNs"])})
        alg = (3 if op == "polymul" else 2) * 1024 * 5760 * 8
        if "avg_ns" in ent:
            ent["alg_bytes"] = alg
            ent["frac_of_8TBps"] = round(alg / ent["avg_ns"] / 8000, 4)
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            v = pmc(f"c4_pmc_{op}_{qb}_{c}", "k_mixed")
            if c in v:
                ent[c + "_KB"] = round(v[c][0], 1)
        if "FETCH_SIZE_KB" in ent and "WRITE_SIZE_KB" in ent:
            ent["hbm_bytes_x2_rule"] = int((2 * ent["FETCH_SIZE_KB"] + ent["WRITE_SIZE_KB"]) * 1024)
            ent["traffic_over_algorithmic_x2_rule"] = round(ent["hbm_bytes_x2_rule"] / alg, 3)
        sqv = pmc(f"c4_sq_{op}_{qb}", "k_mixed")
        ent["sq"] = {k: v[0] for k, v in sorted(sqv.items())}
        if "SQ_INSTS_VALU" in ent["sq"] and "SQ_WAVES" in ent["sq"]:
            ent["valu_instructions_per_wave"] = round(ent["sq"]["SQ_INSTS_VALU"] / ent["sq"]["SQ_WAVES"], 1)
        c4[f"{op}_q{qb}"] = ent
json.dump(c4, open(os.path.join(DST, f"{tag}_generic_c4.json"), "w"), indent=1)
for name in ("microbench_ops.txt", "microbench_bfly2.txt"):
    src = os.path.join(SRC, name)
    if os.path.exists(src):
        open(os.path.join(DST, f"{tag}_{name}"), "w").write(open(src).read())

# 3c. the reference's own index shapes: timing table + trace/counters of the fused poly-mul at m = 14400
rp = os.path.join(SRC, "refparams.txt")
if os.path.exists(rp):
    with open(os.path.join(DST, f"{tag}_refparams.txt"), "w") as fh:
        fh.write("# tools/bench_refparams2.sh: m = 2^e * odd, batch 8192, one modulus just above 2^26 / 2^58; 'fused' = one launch of the vector\n"
                 "# interpreter with ST_POW2 tiles (default), 'split' = LOLHIP_NO_FUSED2=1 (m = 2^k kernels + odd stage program, unfused poly-mul)\n")
        fh.write(open(rp).read())
        ent = {}
        for f in glob.glob(os.path.join(SRC, "ref_kt", "**", "*kernel_stats.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if "k_mixed" in r["Name"]:
                    ent = {"kernel": (re.search(r"k_mixed<[^>]*>", r["Name"]) or [r["Name"][:60]])[0], "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"])}
        alg = 3 * 8192 * 3840 * 8
        fh.write(f"# rocprofv3 --kernel-trace --stats -- tools/bench_kernels m14400 1 8192 polymul 50 26: {json.dumps(ent)}\n")
        f_, w_ = pmc("ref_pmc_FETCH_SIZE", "k_mixed"), pmc("ref_pmc_WRITE_SIZE", "k_mixed")
        if "FETCH_SIZE" in f_ and "WRITE_SIZE" in w_:
            hb = (2 * f_["FETCH_SIZE"][0] + w_["WRITE_SIZE"][0]) * 1024
            fh.write(f"# FETCH_SIZE {f_['FETCH_SIZE'][0]:.1f} KB (x2 rule), WRITE_SIZE {w_['WRITE_SIZE'][0]:.1f} KB per launch: {hb / alg:.3f} x algorithmic ({alg} B)\n")
        sqv = pmc("ref_sq", "k_mixed")
        for k in sorted(sqv):
            fh.write(f"# {k:24s} {sqv[k][0]:.6g}\n")

# 4. pipeline kernels
pl = os.path.join(SRC, "pipelines.jsonl")
if os.path.exists(pl):
    lines = [l for l in open(pl) if l.startswith("{")]
    open(os.path.join(DST, f"{tag}_pipelines.jsonl"), "w").writelines(lines)
print(json.dumps(bench["roofline"]), traffic["traffic_over_algorithmic"])
print(open(os.path.join(DST, f"{tag}_bench_kernel_stats.csv")).read()[:600])
