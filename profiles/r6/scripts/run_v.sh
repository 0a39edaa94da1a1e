#!/bin/bash
# round 6, call V: the driver's own sequence on a second box with the final library (build 1d39f27f7bc23935) + the N = 8192 traffic table of that build
out=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
{
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
t0=$(date +%s)
python bench.py --gpus 1 --steps 20 --warmup 5 > $out/r6v_default.json 2>/dev/null
echo "default bench wall $(( $(date +%s) - t0 )) s"
for c in pemsbay expytky; do python bench.py --config $c --no-cpu-baseline > $out/r6v_$c.json 2>/dev/null; done
python bench.py --config syn8192 --steps 5 --warmup 2 --no-cpu-baseline > $out/r6v_syn8192.json 2>/dev/null
python - <<'PY'
import json, os
o = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/"
d = json.load(open(o + "r6v_default.json"))
print("metrla", d["value"], d["ms_per_step"], d["roofline"]["frac"], "noT", d["value_no_teacher"], d["roofline"].get("traffic_source", {}).get("status"))
for k in ("secondary", "secondary_parity", "syn8192"):
    s = d[k]; r = s.get("roofline", {}); print(" ", k, s.get("dtype"), s.get("value"), s.get("ms_per_step"), r.get("frac"), r.get("shader_clock_mhz"), r.get("frac_of_mfma_peak_at_held_clock"), s.get("skipped"), s.get("error"))
for c in ("pemsbay", "expytky", "syn8192"):
    d = json.load(open(o + f"r6v_{c}.json")); r = d["roofline"]; print(c, d["value"], d["ms_per_step"], r["frac"], r.get("shader_clock_mhz"))
PY
} > $out/r6v.log 2>&1
MCRN_GIT_REV=$MCRN_GIT_REV bash tools/pmc_traffic.sh r6_syn8192 --config syn8192 > $out/r6v_traffic_syn8192.log 2>&1
cat $out/r6v.log; head -3 $out/r6v_traffic_syn8192.log
