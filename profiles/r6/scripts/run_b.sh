#!/bin/bash
# round 6, call B: where the time of the hi/lo K loop goes.  (1) time vs K (1, 2, 3 segments of 1843): slope = the K loop, intercept =
# launch + prologue + epilogue; with and without the output stores.  (2) the loop without one of its streams (MCRN_BF16_ABL: 1 = no
# operand DMA, 2 = fragments read once, 4 = no MFMA), (3) the clock the chip holds under the loop (ABL 8).
cd $GRAFT_REPO_ROOT/tools/kbench
{
for cfg in 3 7 1; do
  for N in 2048 1024; do
    for ns in 1 2 3; do
      echo "-- X3 cfg $cfg N $N nseg $ns"
      X3=1 ./bf16_abl0 7372 $N 1843 $ns nn $cfg 1 20 | tail -1
      X3=1 NO_OUT=1 ./bf16_abl0 7372 $N 1843 $ns nn $cfg 1 20 | tail -1
    done
  done
done
echo "== plain cfg 4 / 3"
for ns in 1 2 3; do
  ./bf16_abl0 7372 2048 1843 $ns nn 4 1 20 | tail -1
  NO_OUT=1 ./bf16_abl0 7372 2048 1843 $ns nn 4 1 20 | tail -1
  CB_ONLY=1 ./bf16_abl0 7372 2048 1843 $ns nn 4 1 20 | tail -1
done
echo "== ablations, X3 cfg 3 (N 2048) and cfg 7 / 1 (N 1024), NO_OUT, 3 segments"
for a in 0 1 2 4; do
  echo "-- ABL $a"
  X3=1 NO_OUT=1 ./bf16_abl$a 7372 2048 1843 3 nn 3 1 20 | tail -1
  X3=1 NO_OUT=1 ./bf16_abl$a 7372 1024 1843 3 nn 7 1 20 | tail -1
  X3=1 NO_OUT=1 ./bf16_abl$a 7372 1024 1843 3 nn 1 1 20 | tail -1
  NO_OUT=1 ./bf16_abl$a 7372 2048 1843 3 nn 4 1 20 | tail -1
done
echo "== clock"
X3=1 ./bf16_abl8 7372 2048 1843 1 nn 3 1 20 | tail -2
X3=1 ./bf16_abl8 7372 2048 1843 3 nn 3 1 20 | tail -2
./bf16_abl8 7372 2048 1843 3 nn 4 1 20 | tail -2
} > $GRAFT_REPO_ROOT/gpurun_out/r6b.log 2>&1
cat $GRAFT_REPO_ROOT/gpurun_out/r6b.log
