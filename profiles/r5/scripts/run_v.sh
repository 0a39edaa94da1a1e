#!/bin/bash
# round 5, call V: final small-graph library (attached ready event + per-cell plane sets, knobs removed): parity; bf16 mode: width of the
# decoder's deferred weight gradients beside the encoder BPTT (they stretch the main queue's kernels 3 - 4x at full width: timeline)
out=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
q() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d.get('value_no_teacher'), d.get('kernel_launches_per_step'))"; }
{
echo "== parity"
timeout 900 python -m pytest tests -m gpu -x -q -k "kernel_variants or model_train_step or golden or trajectory" 2>&1 | tail -3
echo "== A/B decoder weight-gradient width, bf16 mode"
for rep in 1 2; do
for w in 256 96 48 24; do
MCRN_DEC_WG_BF16=$w python bench.py --config expytky --no-secondary --no-cpu-baseline --no-roofline --no-syn 2>/dev/null | q "expytky dec_wg=$w"
done
done
for w in 256 96 48; do
MCRN_DEC_WG_BF16=$w python bench.py --config syn8192 --steps 3 --warmup 1 --no-secondary --no-cpu-baseline --no-roofline --no-regimes 2>/dev/null | q "syn8192 dec_wg=$w"
done
for w in 256 96 48; do
MCRN_DEC_WG_BF16=$w python bench.py --config expytky --precision bf16x3 --no-secondary --no-cpu-baseline --no-roofline --no-regimes --no-syn 2>/dev/null | q "expytky bf16x3 dec_wg=$w"
done
} > $out/r5v.log 2>&1
cat $out/r5v.log
