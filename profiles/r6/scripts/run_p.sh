#!/bin/bash
# round 6, call P: the d-grad's output forms as a template parameter (fp32-only kernels: 195 -> 130 VGPRs at O = 128, a third wave per SIMD) against the
# previous library, alternating, METR-LA and PEMS-BAY (fp32 form) and EXPY-TKY (hoisted form: unchanged code)
out=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
q() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); g=d.get('gemm_roles',{})
print('$1', d['value'], d['ms_per_step'], 'dgrad', g.get('dgrad',{}).get('ms_per_step'), g.get('dgrad',{}).get('avg_us'), 'noT', d.get('value_no_teacher'))"; }
{
for rep in 1 2 3; do
  python bench.py --no-secondary --no-syn --no-cpu-baseline 2>/dev/null | q "metrla new "
  MEGACRN_LIB=$GRAFT_REPO_ROOT/megacrn_amd/libmegacrn_hip_prev.so python bench.py --no-secondary --no-syn --no-cpu-baseline 2>/dev/null | q "metrla prev"
done
for rep in 1 2; do
  python bench.py --config pemsbay --no-cpu-baseline 2>/dev/null | q "pemsbay new "
  MEGACRN_LIB=$GRAFT_REPO_ROOT/megacrn_amd/libmegacrn_hip_prev.so python bench.py --config pemsbay --no-cpu-baseline 2>/dev/null | q "pemsbay prev"
done
python bench.py --config expytky --no-cpu-baseline 2>/dev/null | q "expytky new "
MEGACRN_LIB=$GRAFT_REPO_ROOT/megacrn_amd/libmegacrn_hip_prev.so python bench.py --config expytky --no-cpu-baseline 2>/dev/null | q "expytky prev"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "variants or model_train_step or full_size_metrla" 2>&1 | tail -2
} > $out/r6p.log 2>&1
cat $out/r6p.log
