#!/bin/bash
# round 6, call F: the library with G1L (ping-pong tiles of three and more stages issue every DMA instruction in a load phase) in the model:
# EXPY-TKY in both arithmetics, N = 8192, METR-LA / PEMS-BAY for reference; the tuner's tables
out=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
MCRN_TUNE_LOG=1 python bench.py --config expytky --precision bf16x3 --no-cpu-baseline > $out/r6f_expytky_x3.json 2> $out/r6f_expytky_x3.err
MCRN_TUNE_LOG=1 python bench.py --config expytky --no-cpu-baseline > $out/r6f_expytky.json 2> $out/r6f_expytky.err
MCRN_TUNE_LOG=1 python bench.py --config syn8192 --steps 5 --warmup 2 --no-cpu-baseline > $out/r6f_syn8192.json 2> $out/r6f_syn8192.err
python bench.py --no-secondary --no-syn --no-cpu-baseline > $out/r6f_metrla.json 2> /dev/null
python bench.py --config pemsbay --no-cpu-baseline > $out/r6f_pemsbay.json 2> /dev/null
python - <<'PY'
import json, os
o = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/"
for f in ("r6f_metrla", "r6f_pemsbay", "r6f_expytky_x3", "r6f_expytky", "r6f_syn8192"):
    d = json.load(open(o + f + ".json")); r = d["roofline"]
    print(f, d["value"], d["ms_per_step"], r["frac"], r.get("shader_clock_mhz"), r.get("frac_of_mfma_peak_at_held_clock"), r["avg_launch_us"], d.get("value_no_teacher"), d.get("eval_samples_per_s"))
    for k, v in d.get("gemm_roles", {}).items(): print("    ", k, v["ms_per_step"], v["avg_us"], v["alg_tflops"])
PY
