"""tools/summarize_pipelines_pmc.py <gpurun_out/pipes> <out.json>: per kernel of tools/bench_pipelines.py —
calls and average duration from the kernel trace, mean FETCH_SIZE / WRITE_SIZE per dispatch from the two --pmc passes
(rocprofv3 reports them in KiB on this build? no: in the counter's own unit, which the copy calibration fixes),
and the calibration factors from the 256 MiB copy whose bytes are known."""
import collections, csv, glob, json, re, sys
src, dst = sys.argv[1], sys.argv[2]


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*", "", name)
    return name.replace("lolhip::", "")[:80]


def pmc_mean(d, counter):
    agg = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


stats = {}
for f in glob.glob(src + "/kt/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        stats[short(r["Name"])] = {"calls": int(r["Calls"]), "avg_us": round(float(r["AverageNs"]) / 1e3, 2)}
fetch, write = pmc_mean(src + "/pmc_FETCH_SIZE", "FETCH_SIZE"), pmc_mean(src + "/pmc_WRITE_SIZE", "WRITE_SIZE")
cal_f, cal_w = pmc_mean(src + "/cal_FETCH_SIZE", "FETCH_SIZE"), pmc_mean(src + "/cal_WRITE_SIZE", "WRITE_SIZE")
copy_bytes = 4096 * 8192 * 8
kf = [v for k, v in cal_f.items() if "k_copy16" in k]
kw = [v for k, v in cal_w.items() if "k_copy16" in k]
out = {"calibration": {"copy_bytes_each_way": copy_bytes, "FETCH_SIZE_reported": kf[0] if kf else None, "WRITE_SIZE_reported": kw[0] if kw else None,
                       "bytes_per_FETCH_unit": round(copy_bytes / kf[0], 3) if kf else None, "bytes_per_WRITE_unit": round(copy_bytes / kw[0], 3) if kw else None,
                       "note": "k_copy16 reads and writes 268,435,456 B with 16 B per lane; MI355X_MICROARCH.md: FETCH_SIZE counts a wide streaming read at half"},
       "headline": {}, "kernels": {}}
ff, fw = (copy_bytes / kf[0] if kf else None), (copy_bytes / kw[0] if kw else None)
for tag, d in (("k_pow2<13,2,1,true> 61-bit poly-mul (alg 805306368 B)", "hl"), ("k_pow2_pipe<13,4,false> 27-bit poly-mul (alg 805306368 B)", "pp")):
    f_, w_ = pmc_mean(src + f"/{d}_FETCH_SIZE", "FETCH_SIZE"), pmc_mean(src + f"/{d}_WRITE_SIZE", "WRITE_SIZE")
    fk = [v for k, v in f_.items() if "k_pow2" in k]
    wk = [v for k, v in w_.items() if "k_pow2" in k]
    if fk and wk and ff and fw:
        out["headline"][tag] = {"fetch_bytes": round(fk[0] * ff), "write_bytes": round(wk[0] * fw), "traffic_over_algorithmic": round((fk[0] * ff + wk[0] * fw) / 805306368, 3)}
for k in sorted(set(stats) | set(fetch) | set(write)):
    e = dict(stats.get(k, {}))
    if k in fetch and ff:
        e["fetch_MB_per_launch"] = round(fetch[k] * ff / 1e6, 2)
    if k in write and fw:
        e["write_MB_per_launch"] = round(write[k] * fw / 1e6, 2)
    out["kernels"][k] = e
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out["calibration"]), json.dumps(out["headline"]))
