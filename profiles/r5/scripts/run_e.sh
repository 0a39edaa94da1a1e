#!/bin/bash
# round 5, call E: GRU backward step B fused into the gate call's streaming d-grad: parity + same-call A/B
out=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
q() { python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d.get('gemm_roles',{}); print('$1', d['value'], d['ms_per_step'], d.get('value_no_teacher'), 'dgrad', (r.get('dgrad') or {}).get('avg_us'), 'launches', d.get('kernel_launches_per_step'))"; }
{
echo "== parity (fused default)"
timeout 1500 python -m pytest tests -m gpu -x -q -k "not alternative_paths and not harness" 2>&1 | tail -5
echo "== A/B"
for rep in 1 2 3; do
  MCRN_FUSE_B4=0 python bench.py --no-secondary --no-cpu-baseline 2>/dev/null | q "metrla unfused"
  python bench.py --no-secondary --no-cpu-baseline 2>/dev/null | q "metrla fused"
done
for rep in 1 2; do
  MCRN_FUSE_B4=0 python bench.py --config pemsbay --no-secondary --no-cpu-baseline 2>/dev/null | q "pemsbay unfused"
  python bench.py --config pemsbay --no-secondary --no-cpu-baseline 2>/dev/null | q "pemsbay fused"
  MCRN_FUSE_B4=0 python bench.py --config expytky --no-secondary --no-cpu-baseline 2>/dev/null | q "expytky unfused"
  python bench.py --config expytky --no-secondary --no-cpu-baseline 2>/dev/null | q "expytky fused"
done
} > $out/r5e.log 2>&1
tail -40 $out/r5e.log
