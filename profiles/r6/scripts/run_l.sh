#!/bin/bash
# round 6, call L: the 6 % of call K belong to the accumulator-preload branch of the ring kernel (round 5: SGPRs 84 -> 104 in EVERY instantiation,
# although only the split-0 workgroups of the transposed product take it)?  The same source with the branch / the clock stamps compiled out.
cd $GRAFT_REPO_ROOT/tools/kbench
for rep in 1 2 3; do
  for b in bf16_old bf16_abx_base bf16_abx_nopre bf16_abx_noclk bf16_abx_both; do
    echo "$b: $(CB_ONLY=1 ./$b 7372 1024 1843 1 nn 1 1 40 1 | tail -1)   $(./$b 1843 1024 1843 4 nn 1 4 40 | tail -1)"
  done
done
