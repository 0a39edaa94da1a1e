#!/bin/bash
# round 5, call Z: adjacency gradient of the small graphs, two cells per launch (MCRN_DS_TWO) - parity and A/B
out=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
q() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d.get('value_no_teacher'), d.get('kernel_launches_per_step'))"; }
{
echo "== parity"
timeout 900 python -m pytest tests -m gpu -x -q -k "kernel_variants or model_train_step or golden or trajectory or half_batches or baseline_config_train or full_size_metrla" 2>&1 | tail -3
echo "== A/B"
for rep in 1 2 3; do
for c in metrla pemsbay; do
python bench.py --config $c --no-secondary --no-cpu-baseline --no-roofline --no-syn 2>/dev/null | q "$c two-cells"
MCRN_DS_TWO=0 python bench.py --config $c --no-secondary --no-cpu-baseline --no-roofline --no-syn 2>/dev/null | q "$c per-cell"
done
done
} > $out/r5z.log 2>&1
cat $out/r5z.log
