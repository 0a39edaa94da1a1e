#!/bin/bash
# round 5, call K: parity after the prop2_fwd XCD mapping / final ds4 slab counts, harness, traffic tables, bench of the three small configs
out=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
q() { python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d.get('gemm_roles',{}); print('$1', d['value'], d['ms_per_step'], d.get('value_no_teacher'), d.get('eval_samples_per_s'), 'prop', (r.get('propagate') or {}).get('avg_us'), 'wp', (r.get('weight_pool') or {}).get('avg_us'), 'ds', (r.get('adjacency_grad') or {}).get('avg_us'))"; }
{
(cd tools/kbench && ./prop1_test 207 4352 20 && ./prop1_test 325 4352 20 && ./prop1_test 207 8448 20)
timeout 1200 python -m pytest tests -m gpu -x -q -k "(model_train_step or model_eval or golden or kernel_variants or baseline_config_train or trajectory or large_graph or full_size_metrla or harness) and not bf16_mode" 2>&1 | tail -3
for rep in 1 2; do
python bench.py --no-secondary --no-cpu-baseline 2>/dev/null | q "metrla"
python bench.py --config pemsbay --no-secondary --no-cpu-baseline 2>/dev/null | q "pemsbay"
python bench.py --config expytky --no-secondary --no-cpu-baseline 2>/dev/null | q "expytky"
done
bash tools/pmc_traffic.sh r5k_metrla --config metrla
bash tools/pmc_traffic.sh r5k_pemsbay --config pemsbay
bash tools/pmc_traffic.sh r5k_expytky --config expytky
} > $out/r5k.log 2>&1
tail -90 $out/r5k.log
