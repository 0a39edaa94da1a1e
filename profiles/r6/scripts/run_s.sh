#!/bin/bash
# round 6, call S: half-plane stages in the streaming weight pool at H = 128 with fp32 planes (the METR-LA / PEMS-BAY decoder): 64 -> 32 KB of LDS and
# 160 (+32) -> 136 VGPRs, three workgroups per CU instead of two; against the previous library, alternating
out=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
q() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); g=d.get('gemm_roles',{})
print('$1', d['value'], d['ms_per_step'], 'wp', g.get('weight_pool',{}).get('ms_per_step'), g.get('weight_pool',{}).get('avg_us'), 'noT', d.get('value_no_teacher'), 'eval', d.get('eval_samples_per_s'))"; }
L=$GRAFT_REPO_ROOT/megacrn_amd
{
for rep in 1 2 3; do
  python bench.py --no-secondary --no-syn --no-cpu-baseline 2>/dev/null | q "metrla new "
  MEGACRN_LIB=$L/libmegacrn_hip_prev.so python bench.py --no-secondary --no-syn --no-cpu-baseline 2>/dev/null | q "metrla prev"
done
for rep in 1 2; do
  python bench.py --config pemsbay --no-cpu-baseline 2>/dev/null | q "pemsbay new "
  MEGACRN_LIB=$L/libmegacrn_hip_prev.so python bench.py --config pemsbay --no-cpu-baseline 2>/dev/null | q "pemsbay prev"
done
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "variants or model_train_step or full_size_metrla or cell_golden or baseline_config" 2>&1 | tail -2
} > $out/r6s.log 2>&1
cat $out/r6s.log
