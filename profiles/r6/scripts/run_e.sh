#!/bin/bash
# round 6, call E: every DMA instruction of the ping-pong loop in a LOAD phase (G1L: group 1 runs group 0's schedule one phase later; needs
# NSTAGE >= 3) against round 4's form (group 1's pieces between its MFMAs).  Slots 3 / 5 / 6 / 9 (hi/lo: 16-deep, four stages; plain: 32-deep,
# four stages) and 7; slot 4 (two stages: unchanged code) as the reference.
cd $GRAFT_REPO_ROOT/tools/kbench
{
for b in bf16_g1l bf16_g1l0; do
echo "== $b: X3"
for cfg in 3 4 5 6 7 9; do
  X3=1 ./$b 7372 1024 1843 1 nn $cfg 1 20 | tail -2
  X3=1 ./$b 7372 2048 1843 1 nn $cfg 1 20 | tail -1
  X3=1 ./$b 1843 1024 1843 4 nn $cfg 4 20 | tail -1
  X3=1 ./$b 1843 2048 1843 4 nn $cfg 4 20 | tail -1
  X3=1 ./$b 7372 1843 1024 12 nt $cfg 1 5 | tail -1
done
echo "== $b: plain"
for cfg in 3 4 5 6 7; do
  ./$b 7372 1024 1843 1 nn $cfg 1 20 | tail -2
  ./$b 7372 2048 1843 1 nn $cfg 1 20 | tail -1
  ./$b 1843 2048 1843 4 nn $cfg 2 20 | tail -1
  ./$b 7372 1843 2048 12 nt $cfg 1 5 | tail -1
done
for cfg in 3 4 5; do ./$b 32768 4096 8192 1 nn $cfg 1 5 | tail -1; done
done
echo "== clock, X3 cfg 3 (16-deep, four stages, G1L) and plain cfg 3, 3 segments"
X3=1 ./bf16_g1l8 7372 2048 1843 3 nn 3 1 20 | tail -2
./bf16_g1l8 7372 2048 1843 3 nn 3 1 20 | tail -2
} > $GRAFT_REPO_ROOT/gpurun_out/r6e.log 2>&1
cat $GRAFT_REPO_ROOT/gpurun_out/r6e.log
