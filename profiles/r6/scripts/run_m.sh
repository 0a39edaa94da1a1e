#!/bin/bash
# round 6, call M: the accumulator preload compiled into the ROLE 4 kernels only (default) vs into none (MCRN_BF16_PRELOAD=0 build), in the model;
# EXPY-TKY both arithmetics and N = 8192, alternating libraries inside one call
out=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
q() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; g=d.get('gemm_roles',{})
print('$1', d['value'], d['ms_per_step'], 'fwd', r['frac'], r['avg_launch_us'], r.get('shader_clock_mhz'), 'propT', g.get('propagate_T',{}).get('avg_us'), g.get('propagate_T',{}).get('alg_tflops'), 'prop', g.get('propagate',{}).get('avg_us'))"; }
{
for rep in 1 2; do
  python bench.py --config expytky --no-cpu-baseline 2>/dev/null | q "expytky bf16 roleonly"
  MEGACRN_LIB=$GRAFT_REPO_ROOT/megacrn_amd/libmegacrn_hip_nopre.so python bench.py --config expytky --no-cpu-baseline 2>/dev/null | q "expytky bf16 nopreload"
  python bench.py --config expytky --precision bf16x3 --no-cpu-baseline 2>/dev/null | q "expytky x3 roleonly"
  MEGACRN_LIB=$GRAFT_REPO_ROOT/megacrn_amd/libmegacrn_hip_nopre.so python bench.py --config expytky --precision bf16x3 --no-cpu-baseline 2>/dev/null | q "expytky x3 nopreload"
done
python bench.py --config syn8192 --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | q "syn8192 roleonly"
MEGACRN_LIB=$GRAFT_REPO_ROOT/megacrn_amd/libmegacrn_hip_nopre.so python bench.py --config syn8192 --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | q "syn8192 nopreload"
python bench.py --no-secondary --no-syn --no-cpu-baseline 2>/dev/null | q "metrla roleonly"
} > $out/r6m.log 2>&1
cat $out/r6m.log
