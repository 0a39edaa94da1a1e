#!/bin/bash
# round 6, call K: did the encoder tile of the N = 1843 forward product regress in the KERNEL since round 4 (review: 29.2 -> 31.6 us in the steady
# tables of rounds 4 / 5)?  Round 4's gemm_bf16.h (commit 20405a2) and this round's, same call, same box, same shapes; interleaved repetitions.
cd $GRAFT_REPO_ROOT/tools/kbench
for rep in 1 2 3; do
  for b in bf16_old bf16_gemm_test; do
    echo "== $b (rep $rep)"
    for cfg in 1 13 9; do CB_ONLY=1 ./$b 7372 1024 1843 1 nn $cfg 1 40 1 | tail -1; done
    CB_ONLY=1 ./$b 7372 2048 1843 1 nn 4 1 40 1 | tail -1
    ./$b 1843 2048 1843 4 nn 4 2 40 | tail -1
  done
done
