#!/bin/bash
# round 5, last call: the driver's own sequence on a fresh box - smoke(), the default bench line, the per-config lines (second box for the spread)
out=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
{
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
t0=$(date +%s)
python bench.py > $out/r5_final_default.json 2>/dev/null
echo "default bench wall $(( $(date +%s) - t0 )) s"
for c in pemsbay expytky; do python bench.py --config $c --no-cpu-baseline > $out/r5_final_$c.json 2>/dev/null; done
python bench.py --config syn8192 --steps 5 --warmup 2 --no-cpu-baseline > $out/r5_final_syn8192.json 2>/dev/null
python - <<'PY'
import json, os
o = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/"
d = json.load(open(o + "r5_final_default.json"))
print("metrla", d["value"], d["ms_per_step"], d["roofline"]["frac"], "noT", d["value_no_teacher"])
for k in ("secondary", "secondary_parity", "syn8192"):
    s = d[k]; print(" ", k, s["dtype"], s["value"], s["ms_per_step"], s["roofline"]["frac"])
for c in ("pemsbay", "expytky", "syn8192"):
    d = json.load(open(o + f"r5_final_{c}.json")); print(c, d["value"], d["ms_per_step"], d["roofline"]["frac"])
PY
} > $out/r5_final.log 2>&1
cat $out/r5_final.log
