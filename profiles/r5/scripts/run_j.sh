#!/bin/bash
# round 5, call J: where the bytes of a METR-LA / PEMS-BAY step go now (steady-step PMC table) + kernel table + ds4 slab sweep
out=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
q() { python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d.get('gemm_roles',{}); print('$1', d['value'], d['ms_per_step'], d.get('value_no_teacher'), 'ds', (r.get('adjacency_grad') or {}).get('avg_us'))"; }
{
bash tools/pmc_traffic.sh r5j_metrla --config metrla
bash tools/pmc_traffic.sh r5j_pemsbay --config pemsbay
bash tools/prof_stats.sh r5j_metrla --config metrla --no-secondary --no-syn | head -45
for n in 12 16 20; do
  MCRN_NSLAB_S4=$n python bench.py --no-secondary --no-cpu-baseline 2>/dev/null | q "metrla ds4-$n"
  MCRN_NSLAB_S4=$n python bench.py --config pemsbay --no-secondary --no-cpu-baseline 2>/dev/null | q "pemsbay ds4-$n"
done
} > $out/r5j.log 2>&1
tail -90 $out/r5j.log
