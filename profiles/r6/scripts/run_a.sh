#!/bin/bash
# round 6, call A: the fused hi/lo K tile (gemm_bf16.h X3: A_hi, A_lo, B_hi, B_lo fetched once per K tile, three MFMA blocks from LDS)
#   1. stand-alone harness: every hi/lo tile slot on the EXPY-TKY products (forward enc / dec, transposed, adjacency gradient) + ragged shapes
#   2. the x3r parity cases of the GPU suite
#   3. bench: EXPY-TKY in bf16x3 (secondary_parity's shape) with the tuner's table
out=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT/tools/kbench
{
echo "== harness, X3 =="
for cfg in 0 1 2 3 6 7 9 13 14 15; do
  X3=1 ./bf16_gemm_test 7372 1024 1843 1 nn $cfg 1 20
  X3=1 ./bf16_gemm_test 7372 2048 1843 1 nn $cfg 1 20
  X3=1 ./bf16_gemm_test 1843 1024 1843 4 nn $cfg 2 20
  X3=1 ./bf16_gemm_test 1843 2048 1843 4 nn $cfg 2 20
  X3=1 ./bf16_gemm_test 7372 1843 1024 12 nt $cfg 1 5
done
echo "== ragged =="
X3=1 ./bf16_gemm_test 300 200 88 3 nt 9 2 5
X3=1 ./bf16_gemm_test 333 136 77 2 nn 9 1 5 1
X3=1 ./bf16_gemm_test 300 200 88 3 nt 0 2 5
X3=1 ./bf16_gemm_test 333 136 77 2 nn 13 1 5 1
X3=1 ./bf16_gemm_test 1000 520 200 1 nn 3 1 5 1
X3=1 ./bf16_gemm_test 700 333 200 2 nt 6 1 5 1
X3=1 ./bf16_gemm_test 700 333 200 2 nt 2 3 5
X3=1 ./bf16_gemm_test 1000 520 40 1 nn 14 1 5 1
echo "== plain bf16 regression =="
./bf16_gemm_test 7372 1024 1843 1 nn 13 1 20
./bf16_gemm_test 7372 2048 1843 1 nn 4 1 20
./bf16_gemm_test 1843 2048 1843 4 nn 4 2 20
./bf16_gemm_test 333 136 77 2 nn 9 1 5 1
} > $out/r6a_harness.log 2>&1
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "variants or baseline_config or half_batches or full_size" > $out/r6a_tests.log 2>&1
MCRN_TUNE_LOG=1 python bench.py --config expytky --precision bf16x3 --no-cpu-baseline > $out/r6a_expytky_x3.json 2> $out/r6a_expytky_x3.err
tail -5 $out/r6a_tests.log
grep -c OK $out/r6a_harness.log; grep BAD $out/r6a_harness.log
python - <<'PY'
import json, os
d = json.load(open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r6a_expytky_x3.json"))
print("expytky x3", d["value"], d["ms_per_step"], d["roofline"])
PY
