#!/bin/bash
# usage (GPU box, from the repo root): bash tools/kbench/run_bf16.sh > gpurun_out/bf16_kbench.txt
cd tools/kbench
./bf16_gemm_test probe | tail -3
for cfg in 0 1 2 3; do
  ./bf16_gemm_test 7372 1152 1843 1 nn $cfg 1 20
  ./bf16_gemm_test 7372 2176 1843 1 nn $cfg 1 20
done
./bf16_gemm_test 1843 2176 1843 4 nn 0 1 20
./bf16_gemm_test 1843 2176 1843 4 nn 0 4 20
./bf16_gemm_test 7372 1843 2176 1 nt 0 1 20
./bf16_gemm_test 7372 1843 2176 1 nt 1 1 20
./bf16_gemm_test 7372 1843 2176 6 nt 3 1 10
./bf16_gemm_test 300 200 88 3 nt 0 2 5
./bf16_gemm_test 333 136 77 2 nn 0 1 5
./bf16_gemm_test 4096 4096 4096 1 nn 0 1 10
./bf16_gemm_test 4096 4096 4096 1 nn 3 1 10
./bf16_gemm_test 4096 4096 4096 1 nt 3 1 10
./bf16_gemm_test 32768 4224 8192 1 nn 3 1 5
