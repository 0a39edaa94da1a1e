#!/bin/bash
# round 5, call B: bf16 partial planes of the split transposed propagation (A/B in one call) + the new bench line
out=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
q() { python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d.get('gemm_roles',{}); print('$1', d['value'], d['ms_per_step'], d.get('value_no_teacher'), 'propT', r.get('propagate_T'), 'prop', r.get('propagate'))"; }
{
echo "== parity: bf16 mode with packed bf16 partial planes (default) and with fp32 partial planes"
timeout 1200 python -m pytest tests -m gpu -x -q -k "bf16_mode and not alternative_paths" 2>&1 | tail -4
MCRN_BF16_PARTIALS=0 timeout 900 python -m pytest tests -m gpu -x -q -k "bf16_mode_train and 1843" 2>&1 | tail -3
echo "== ycov_dim=5 under forced hoisting"
MCRN_HOIST_FWD=2 timeout 600 python -m pytest tests -m gpu -x -q -k "kernel_variants" 2>&1 | tail -3
echo "== A/B EXPY-TKY"
for rep in 1 2; do
  MCRN_BF16_PARTIALS=0 python bench.py --config expytky --no-secondary --no-cpu-baseline 2>/dev/null | q "expytky fp32-partials"
  MCRN_TUNE_LOG=1 python bench.py --config expytky --no-secondary --no-cpu-baseline 2>$out/r5b_tune_$rep.log | q "expytky bf16-partials"
done
grep "role 4" $out/r5b_tune_1.log | tail -4
MCRN_BF16_PARTIALS=0 python bench.py --config syn8192 --steps 3 --warmup 1 --no-secondary --no-cpu-baseline --no-regimes 2>/dev/null | q "syn fp32-partials"
python bench.py --config syn8192 --steps 3 --warmup 1 --no-secondary --no-cpu-baseline --no-regimes --save-tiles $out/tiles_syn8192_B32_bf16.json 2>/dev/null | q "syn bf16-partials"
echo "== the default line (timed)"
/usr/bin/time -v python bench.py --steps 20 --warmup 5 > $out/r5b_bench_default.json 2> $out/r5b_bench_default.err
grep "Elapsed" $out/r5b_bench_default.err
python - <<'PY'
import json, os
d = json.load(open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r5b_bench_default.json"))
print("headline", d["value"], d["ms_per_step"])
for k in ("secondary", "secondary_parity", "syn8192"):
    s = d.get(k)
    if s: print(k, {kk: s.get(kk) for kk in ("value", "ms_per_step", "dtype", "error", "tile_table_cached")}, (s.get("roofline") or {}).get("frac"), (s.get("roofline") or {}).get("avg_launch_us"))
PY
} > $out/r5b.log 2>&1
tail -40 $out/r5b.log
