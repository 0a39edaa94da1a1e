#!/bin/bash
# round 5, call C: full GPU suite after the clean-up (timed), partial-plane modes A/B with their errors, the default bench line
out=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
q() { python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d.get('gemm_roles',{}); print('$1', d['value'], d['ms_per_step'], d.get('value_no_teacher'), 'propT', (r.get('propagate_T') or {}).get('avg_us'), (r.get('propagate_T') or {}).get('alg_tflops'), 'prop', (r.get('propagate') or {}).get('alg_tflops'))"; }
{
echo "== full GPU suite"
t0=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
echo "suite seconds: $(( $(date +%s) - t0 ))"
echo "== partial-plane modes: errors of the bf16 mode (EXPY geometry B=8) per mode"
for m in 0 1 2 3; do
  echo "-- MCRN_BF16_PARTIALS=$m"
  MCRN_BF16_PARTIALS=$m timeout 600 python -m pytest tests -m gpu -x -q -s -k "bf16_mode_train and 1843-8" 2>&1 | grep -E "worst errors|passed|failed"
done
echo "== A/B EXPY-TKY per mode"
for rep in 1 2; do for m in 0 1 2 3; do
  MCRN_BF16_PARTIALS=$m python bench.py --config expytky --no-secondary --no-cpu-baseline 2>/dev/null | q "expytky partials=$m"
done; done
echo "== the default line (timed)"
t0=$(date +%s)
python bench.py --steps 20 --warmup 5 > $out/r5c_bench_default.json 2> $out/r5c_bench_default.err
echo "bench seconds: $(( $(date +%s) - t0 ))"
python - <<'PY'
import json, os
d = json.load(open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r5c_bench_default.json"))
print("headline", d["value"], d["ms_per_step"], d["roofline"]["frac"])
for k in ("secondary", "secondary_parity", "syn8192"):
    s = d.get(k)
    if s: print(k, {kk: s.get(kk) for kk in ("value", "ms_per_step", "dtype", "error", "tile_table_cached")}, (s.get("roofline") or {}).get("frac"), (s.get("roofline") or {}).get("avg_launch_us"))
PY
} > $out/r5c.log 2>&1
tail -60 $out/r5c.log
