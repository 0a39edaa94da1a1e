#!/bin/bash
# round 5, call D: rotated K walk of the bf16 GEMM (harness A/B), b4-skip probe at METR-LA, K-rot in the model
out=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
q() { python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d.get('gemm_roles',{}); print('$1', d['value'], d['ms_per_step'], d.get('value_no_teacher'), 'propT', (r.get('propagate_T') or {}).get('avg_us'), (r.get('propagate_T') or {}).get('alg_tflops'), 'prop', (r.get('propagate') or {}).get('avg_us'), (r.get('propagate') or {}).get('alg_tflops'))"; }
{
echo "== bf16 GEMM harness: rotated K walk (KROT=1) vs in-order, us per launch"
cd tools/kbench
for krot in 0 1 0 1; do
  echo "-- KROT=$krot"
  KROT=$krot CB_ONLY=1 ./bf16_gemm_test 7372 1024 1843 1 nn 1 1 40 | tail -1      # encoder forward propagation, 256 x 128 ring
  KROT=$krot CB_ONLY=1 ./bf16_gemm_test 7372 1024 1843 1 nn 13 1 40 | tail -1     # ... four-wave 256 x 128
  KROT=$krot CB_ONLY=1 ./bf16_gemm_test 7372 2048 1843 1 nn 4 1 40 | tail -1      # decoder forward, 256 x 256 ping-pong
  KROT=$krot ./bf16_gemm_test 1843 1024 1843 4 nn 1 4 40 | tail -1                # encoder transposed, 4 splits
  KROT=$krot ./bf16_gemm_test 1843 2048 1843 4 nn 1 4 40 | tail -1                # decoder transposed, 4 splits
  KROT=$krot ./bf16_gemm_test 7372 1843 1024 12 nt 4 1 10 | tail -1               # adjacency gradient of the encoder stack
  KROT=$krot CB_ONLY=1 ./bf16_gemm_test 32768 4096 8192 1 nn 4 1 6 | tail -1      # N = 8192 forward
done
cd $GRAFT_REPO_ROOT
echo "== K rotation in the model (EXPY-TKY, SYN)"
for rep in 1 2; do
  python bench.py --config expytky --no-secondary --no-cpu-baseline 2>/dev/null | q "expytky krot=0"
  MCRN_BF16_KROT=1 python bench.py --config expytky --no-secondary --no-cpu-baseline 2>/dev/null | q "expytky krot=1"
done
python bench.py --config syn8192 --steps 3 --warmup 1 --no-secondary --no-cpu-baseline --no-regimes 2>/dev/null | q "syn krot=0"
MCRN_BF16_KROT=1 python bench.py --config syn8192 --steps 3 --warmup 1 --no-secondary --no-cpu-baseline --no-regimes 2>/dev/null | q "syn krot=1"
echo "== METR-LA / PEMS-BAY: bound of folding k_cell_bwd_b4 away (MCRN_DBG_SKIP_B4: wrong gradients, timing only)"
for rep in 1 2; do
  python bench.py --no-secondary --no-cpu-baseline --no-roofline 2>/dev/null | q "metrla default"
  MCRN_DBG_SKIP_B4=1 python bench.py --no-secondary --no-cpu-baseline --no-roofline 2>/dev/null | q "metrla skip-b4"
done
python bench.py --config pemsbay --no-secondary --no-cpu-baseline --no-roofline 2>/dev/null | q "pemsbay default"
MCRN_DBG_SKIP_B4=1 python bench.py --config pemsbay --no-secondary --no-cpu-baseline --no-roofline 2>/dev/null | q "pemsbay skip-b4"
} > $out/r5d.log 2>&1
tail -70 $out/r5d.log
