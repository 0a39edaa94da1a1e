#!/bin/bash
# stream-K configurations (10-12) against their ping-pong siblings (4, 9, 8) on the EXPY-TKY / SYN-8192 products
cd tools/kbench
for cfg in ${CFGS:-4 10 9 11 8 12}; do
  ./bf16_gemm_test 7372 1152 1843 1 nn $cfg 1 20
  ./bf16_gemm_test 7372 2176 1843 1 nn $cfg 1 20
  ./bf16_gemm_test 1843 1152 1843 4 nn $cfg 1 20
  ./bf16_gemm_test 1843 2176 1843 4 nn $cfg 1 20
  ./bf16_gemm_test 7372 1843 2176 12 nt $cfg 1 5
  ./bf16_gemm_test 4096 4096 4096 1 nn $cfg 1 10
done
for cfg in 10 11 12; do
  ./bf16_gemm_test 300 200 88 3 nt $cfg 1 5
  ./bf16_gemm_test 333 136 77 2 nn $cfg 1 5 1
  ./bf16_gemm_test 1000 520 200 1 nn $cfg 1 5 1
  ./bf16_gemm_test 700 333 200 2 nt $cfg 1 5 1
  ./bf16_gemm_test 2500 2000 64 1 nn $cfg 1 5 1
done
./bf16_gemm_test 32768 4224 8192 1 nn 10 1 5
./bf16_gemm_test 8192 4224 8192 4 nn 10 1 5
