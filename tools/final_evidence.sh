#!/bin/bash
# Runs ON THE GPU BOX: the round's closing evidence in one call — full gpu test suite, the driver's own bench command plain and
# under rocprofv3 (tools/profile_bench.sh), kernel statistics + FETCH/WRITE_SIZE of the pipeline kernels (tools/pmc_pipelines.sh).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout -k 10 600 python3 -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests_final.log 2>&1; tail -3 gpurun_out/gpu_tests_final.log
tools/profile_bench.sh > gpurun_out/profile_bench.log 2>&1; head -3 gpurun_out/benchprof/kernel_stats.csv
tools/pmc_pipelines.sh > gpurun_out/pmc_pipelines.log 2>&1
python3 tools/summarize_pipelines_pmc.py gpurun_out/pipes gpurun_out/r03_pipelines_pmc.json | cut -c1-400
cp gpurun_out/pipes/kt.jsonl gpurun_out/r03_pipelines.jsonl
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/benchprof/bench.json"))
print(d["ms_per_step"], d["value"], d["roofline"]["frac"], d["parity_sample_ok"])
s = d["secondary"]
for k, v in s.items():
    if isinstance(v, dict) and "ms" in v:
        print(k, v["ms"], v["frac"], v.get("vs_copy"))
print(s["dropin_c1"])
print(d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
PY
