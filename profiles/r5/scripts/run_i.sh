#!/bin/bash
# round 5, call I: four-block adjacency gradient (ds4) + no d1t write-back: parity, harness, same-call A/B
out=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
q() { python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d.get('gemm_roles',{}); print('$1', d['value'], d['ms_per_step'], d.get('value_no_teacher'), 'ds', (r.get('adjacency_grad') or {}).get('avg_us'), 'propT', (r.get('propagate_T') or {}).get('avg_us'), 'launches', d.get('kernel_launches_per_step'))"; }
{
echo "== harness"
(cd tools/kbench && ./prop1_test 207 4352 20 && ./prop1_test 325 4352 20 && ./prop1_test 207 8448 20)
echo "== parity (ds4 default)"
timeout 1200 python -m pytest tests -m gpu -x -q -k "(model_train_step or golden or kernel_variants or baseline_config_train or trajectory or large_graph or full_size_metrla or strong_scaling or full_batch_backward or alternative_paths) and not bf16_mode" 2>&1 | tail -4
echo "== A/B"
for rep in 1 2; do
  MCRN_DS4=0 python bench.py --no-secondary --no-cpu-baseline 2>/dev/null | q "metrla ds2"
  python bench.py --no-secondary --no-cpu-baseline 2>/dev/null | q "metrla ds4-16"
  MCRN_NSLAB_S4=32 python bench.py --no-secondary --no-cpu-baseline 2>/dev/null | q "metrla ds4-32"
  MCRN_NSLAB_S4=24 python bench.py --no-secondary --no-cpu-baseline 2>/dev/null | q "metrla ds4-24"
done
for rep in 1 2; do
  MCRN_DS4=0 python bench.py --config pemsbay --no-secondary --no-cpu-baseline 2>/dev/null | q "pemsbay ds2"
  python bench.py --config pemsbay --no-secondary --no-cpu-baseline 2>/dev/null | q "pemsbay ds4-16"
  MCRN_NSLAB_S4=32 python bench.py --config pemsbay --no-secondary --no-cpu-baseline 2>/dev/null | q "pemsbay ds4-32"
done
} > $out/r5i.log 2>&1
tail -50 $out/r5i.log
