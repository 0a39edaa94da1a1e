#!/bin/bash
# round 6, call W: the GPU suite of the final tree (adds the weight-gradient harness report, tests/test_gpu_parity.py) + smoke + the default bench line;
# library unchanged (build 1d39f27f7bc23935)
out=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
t0=$(date +%s)
python -m pytest tests -m gpu -x -q -s -k "harness" > $out/r6w_harness.log 2>&1
python -m pytest tests -m gpu -x -q > $out/r6w_tests.log 2>&1
echo "suite wall $(( $(date +%s) - t0 )) s" >> $out/r6w_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 > $out/r6w.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $out/r6w_default.json 2>/dev/null
python - >> $out/r6w.log <<'PY'
import json, os
o = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/"
d = json.load(open(o + "r6w_default.json"))
print("metrla", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("traffic_source", {}).get("status"))
for k in ("secondary", "secondary_parity", "syn8192"):
    s = d[k]; r = s.get("roofline", {}); print(" ", k, s.get("value"), s.get("ms_per_step"), r.get("frac"), r.get("shader_clock_mhz"), s.get("skipped"), s.get("error"))
PY
grep -h "interference report\|passed\|failed" $out/r6w_harness.log | tail -4; tail -3 $out/r6w_tests.log; cat $out/r6w.log
