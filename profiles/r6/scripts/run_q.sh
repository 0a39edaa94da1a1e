#!/bin/bash
# round 6, call Q: the d-grad's A bounce buffer in 64-column chunks (LDS 68 -> 35 KB at O = 128: three workgroups per CU instead of two) on top of the
# output-form template (prev2) and against the library before both (prev); alternating, METR-LA / PEMS-BAY / EXPY-TKY / N = 8192
out=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
q() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); g=d.get('gemm_roles',{})
print('$1', d['value'], d['ms_per_step'], 'dgrad', g.get('dgrad',{}).get('ms_per_step'), g.get('dgrad',{}).get('avg_us'), 'noT', d.get('value_no_teacher'))"; }
L=$GRAFT_REPO_ROOT/megacrn_amd
{
for rep in 1 2 3; do
  python bench.py --no-secondary --no-syn --no-cpu-baseline 2>/dev/null | q "metrla new  "
  MEGACRN_LIB=$L/libmegacrn_hip_prev2.so python bench.py --no-secondary --no-syn --no-cpu-baseline 2>/dev/null | q "metrla prev2"
  MEGACRN_LIB=$L/libmegacrn_hip_prev.so python bench.py --no-secondary --no-syn --no-cpu-baseline 2>/dev/null | q "metrla prev "
done
for rep in 1 2; do
  python bench.py --config pemsbay --no-cpu-baseline 2>/dev/null | q "pemsbay new  "
  MEGACRN_LIB=$L/libmegacrn_hip_prev2.so python bench.py --config pemsbay --no-cpu-baseline 2>/dev/null | q "pemsbay prev2"
done
for rep in 1 2; do
  python bench.py --config expytky --no-cpu-baseline 2>/dev/null | q "expytky new  "
  MEGACRN_LIB=$L/libmegacrn_hip_prev2.so python bench.py --config expytky --no-cpu-baseline 2>/dev/null | q "expytky prev2"
done
python bench.py --config syn8192 --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | q "syn8192 new  "
MEGACRN_LIB=$L/libmegacrn_hip_prev2.so python bench.py --config syn8192 --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | q "syn8192 prev2"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "variants or model_train_step or full_size_metrla or agcn_golden or cell_golden" 2>&1 | tail -2
} > $out/r6q.log 2>&1
cat $out/r6q.log
