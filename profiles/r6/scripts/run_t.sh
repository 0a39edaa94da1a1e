#!/bin/bash
# round 6, call T: x3r - the GRU epilogues of the streaming weight pool emit the packed hi/lo operand of the next propagation product (the pack pass
# runs for the first cell of a stack only: 24 -> 2 launches of k_pack_cols_bf16 per forward); against the previous library, alternating; the x3r parity cases
out=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
q() { python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$1', d['value'], d['ms_per_step'], 'launches', d.get('kernel_launches_per_step'), 'noT', d.get('value_no_teacher'), 'eval', d.get('eval_samples_per_s'))"; }
L=$GRAFT_REPO_ROOT/megacrn_amd
{
for rep in 1 2 3; do
  python bench.py --config expytky --precision bf16x3 --no-cpu-baseline --no-roofline 2>/dev/null | q "expytky x3 new "
  MEGACRN_LIB=$L/libmegacrn_hip_prev.so python bench.py --config expytky --precision bf16x3 --no-cpu-baseline --no-roofline 2>/dev/null | q "expytky x3 prev"
done
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "variants or baseline_config or half_batches or full_size or trajectory_tracks" 2>&1 | tail -2
} > $out/r6t.log 2>&1
cat $out/r6t.log
