#!/bin/bash
# round 6, call U: the full cycle on the final library (+ x3r epilogue-emitted operands): smoke(), the GPU suite, tools/measure_round.sh
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6u_smoke.log 2>&1; tail -1 gpurun_out/r6u_smoke.log
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r6u_tests.log 2>&1; tail -3 gpurun_out/r6u_tests.log
bash tools/measure_round.sh r6 profiles/r6 > gpurun_out/r6u_measure.log 2>&1; echo "measure_round rc $?"
python - <<'PY'
import json, os
o = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/"
d = json.loads(open(o + "r6_bench_metrla.json").read().strip().splitlines()[-1])
print("metrla", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("traffic"), d.get("step_fabric_gb"), d["roofline"].get("traffic_source", {}).get("status"))
for k in ("secondary", "secondary_parity", "syn8192"):
    s = d.get(k, {}); r = s.get("roofline", {})
    print(" ", k, s.get("dtype"), s.get("value"), s.get("ms_per_step"), r.get("frac"), r.get("shader_clock_mhz"), r.get("frac_of_mfma_peak_at_held_clock"), s.get("tile_table_cached"), s.get("skipped"), s.get("error"))
for c in ("pemsbay", "expytky", "syn8192"):
    d = json.loads(open(o + f"r6_bench_{c}.json").read().strip().splitlines()[-1]); print(c, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("shader_clock_mhz"), d.get("step_fabric_gb"))
PY
