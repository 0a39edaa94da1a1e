#!/bin/bash
# round 5, call F: fused GRU-backward loader (batched loads) A/B + per-XCD K rotation harness + parity subset
out=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
q() { python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d.get('gemm_roles',{}); print('$1', d['value'], d['ms_per_step'], d.get('value_no_teacher'), 'dgrad', (r.get('dgrad') or {}).get('avg_us'), 'launches', d.get('kernel_launches_per_step'))"; }
{
echo "== parity subset (fused default)"
timeout 900 python -m pytest tests -m gpu -x -q -k "model_train_step or golden or kernel_variants or baseline_config_train or trajectory or (bf16_mode_train and 1843)" 2>&1 | tail -3
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
echo "== bf16 GEMM harness: K walk rotated PER XCD (lock-step inside an XCD, staggered across XCDs)"
cd tools/kbench
for krot in 0 1 0 1; do
  echo "-- KROT=$krot"
  KROT=$krot CB_ONLY=1 ./bf16_gemm_test_xrot 7372 1024 1843 1 nn 1 1 40 | tail -1
  KROT=$krot CB_ONLY=1 ./bf16_gemm_test_xrot 7372 2048 1843 1 nn 4 1 40 | tail -1
  KROT=$krot ./bf16_gemm_test_xrot 1843 1024 1843 4 nn 1 4 40 | tail -1
  KROT=$krot CB_ONLY=1 ./bf16_gemm_test_xrot 32768 4096 8192 1 nn 4 1 6 | tail -1
done
} > $out/r5f.log 2>&1
tail -50 $out/r5f.log
