#!/bin/bash
# round 5, call Y: helper-stream events - no `done` events when every cell owns its plane sets (next), + events without the system-scope
# fence (next2); A/B against the committed library
out=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
q() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d.get('value_no_teacher'), d.get('kernel_launches_per_step'))"; }
L=$GRAFT_REPO_ROOT/megacrn_amd
{
echo "== parity (next2)"
MEGACRN_LIB=$L/libmegacrn_hip_next2.so timeout 900 python -m pytest tests -m gpu -x -q -k "kernel_variants or model_train_step or golden or trajectory or half_batches or bf16_mode_train" 2>&1 | tail -3
echo "== A/B"
for rep in 1 2 3; do
for c in metrla pemsbay expytky; do
python bench.py --config $c --no-secondary --no-cpu-baseline --no-roofline --no-syn --no-regimes 2>/dev/null | q "$c committed"
MEGACRN_LIB=$L/libmegacrn_hip_next.so python bench.py --config $c --no-secondary --no-cpu-baseline --no-roofline --no-syn --no-regimes 2>/dev/null | q "$c next"
MEGACRN_LIB=$L/libmegacrn_hip_next2.so python bench.py --config $c --no-secondary --no-cpu-baseline --no-roofline --no-syn --no-regimes 2>/dev/null | q "$c next2"
done
done
echo "== timeline (next2)"
MEGACRN_LIB=$L/libmegacrn_hip_next2.so bash tools/prof_stats.sh r5y_metrla --config metrla --no-secondary --no-syn > /dev/null 2>&1
sed -n 330,352p $out/r5y_metrla_timeline.txt | cut -c1-100
} > $out/r5y.log 2>&1
tail -70 $out/r5y.log
