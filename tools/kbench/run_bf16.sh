#!/bin/bash
# usage (GPU box, from the repo root): bash tools/kbench/run_bf16.sh > gpurun_out/bf16_kbench.txt
cd tools/kbench
for cfg in ${CFGS:-4 8 9}; do
  ./bf16_gemm_test 7372 1152 1843 1 nn $cfg 1 20
  ./bf16_gemm_test 7372 2176 1843 1 nn $cfg 1 20
  ./bf16_gemm_test 1843 2176 1843 4 nn $cfg 2 20
  ./bf16_gemm_test 7372 1843 2176 12 nt $cfg 1 5
  ./bf16_gemm_test 4096 4096 4096 1 nn $cfg 1 10
  ./bf16_gemm_test 4096 4096 4096 1 nt $cfg 1 10
done
./bf16_gemm_test 300 200 88 3 nt 8 2 5
./bf16_gemm_test 333 136 77 2 nn 9 1 5 1
./bf16_gemm_test 300 200 88 3 nt 9 2 5
./bf16_gemm_test 333 136 77 2 nn 9 1 5 1
./bf16_gemm_test 1000 520 200 1 nn 8 1 5 1
./bf16_gemm_test 700 333 200 2 nt 9 1 5 1
./bf16_gemm_test 32768 4224 8192 1 nn 4 1 5
