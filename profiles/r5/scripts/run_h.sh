#!/bin/bash
# round 5, call H: full GPU suite on the cleaned library (timed) + default bench line + tile table of the N = 8192 shape
out=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
{
echo "== full GPU suite"
t0=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
echo "suite seconds: $(( $(date +%s) - t0 ))"
echo "== smoke"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "== tile table of SYN-8192 (B = 32, bf16)"
python bench.py --config syn8192 --steps 3 --warmup 1 --no-secondary --no-cpu-baseline --no-regimes --save-tiles $out/tiles_syn8192_B32_bf16.json > $out/r5h_bench_syn8192.json 2>/dev/null
python -c "import json; d=json.load(open('$out/r5h_bench_syn8192.json')); print('syn8192', d['value'], d['ms_per_step'], d['roofline']['frac'])"
mkdir -p profiles/tiles && cp $out/tiles_syn8192_B32_bf16.json profiles/tiles/syn8192_B32_bf16.json
echo "== the default line (timed), with the tile table in place"
t0=$(date +%s)
python bench.py --steps 20 --warmup 5 > $out/r5h_bench_default.json 2> $out/r5h_bench_default.err
echo "bench seconds: $(( $(date +%s) - t0 ))"
python - <<'PY'
import json, os
d = json.load(open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r5h_bench_default.json"))
print("headline", d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("step_fabric_gb"))
for k in ("secondary", "secondary_parity", "syn8192"):
    s = d.get(k)
    if s: print(k, {kk: s.get(kk) for kk in ("value", "ms_per_step", "dtype", "error", "tile_table_cached", "skipped")}, (s.get("roofline") or {}).get("frac"), (s.get("roofline") or {}).get("avg_launch_us"))
PY
} > $out/r5h.log 2>&1
tail -40 $out/r5h.log
