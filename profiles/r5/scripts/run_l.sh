#!/bin/bash
# round 5, call L: transposed propagation of the bf16 mode with the accumulators preloaded from plane 0 (cin_pre): parity + A/B
out=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
q() { python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d.get('gemm_roles',{}); print('$1', d['value'], d['ms_per_step'], d.get('value_no_teacher'), 'propT', (r.get('propagate_T') or {}).get('avg_us'), (r.get('propagate_T') or {}).get('alg_tflops'), 'prop', (r.get('propagate') or {}).get('alg_tflops'))"; }
{
echo "== parity: bf16 mode"
timeout 1500 python -m pytest tests -m gpu -x -q -s -k "bf16_mode or (full_batch_backward and bf16)" 2>&1 | grep -E "worst errors|additivity|passed|failed|FAILED" | head -30
echo "== A/B"
for rep in 1 2 3; do
  MCRN_BF16_CIN_PRE=0 python bench.py --config expytky --no-secondary --no-cpu-baseline 2>/dev/null | q "expytky rmw-epilogue"
  python bench.py --config expytky --no-secondary --no-cpu-baseline 2>/dev/null | q "expytky preload"
done
MCRN_BF16_CIN_PRE=0 python bench.py --config syn8192 --steps 3 --warmup 1 --no-secondary --no-cpu-baseline --no-regimes 2>/dev/null | q "syn rmw-epilogue"
python bench.py --config syn8192 --steps 3 --warmup 1 --no-secondary --no-cpu-baseline --no-regimes 2>/dev/null | q "syn preload"
} > $out/r5l.log 2>&1
tail -50 $out/r5l.log
