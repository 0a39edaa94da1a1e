#!/bin/bash
# round 6, call C: (1) persistent multi-cell propagation prototype vs the launched chain (kill criterion 15 %), (2) the shader clock the chip
# holds under the GEMM K loops with and without their DMA / MFMA streams, the "G0 issues first" A/B, (3) the full GPU suite on the new library
out=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT/tools/kbench
{
echo "== sysfs clocks readable?"
ls /sys/class/drm/ 2>&1 | head -5
for f in /sys/class/drm/card*/device/pp_dpm_sclk; do echo $f; cat $f 2>&1 | head -12; done
for f in /sys/class/drm/card*/device/hwmon/hwmon*/freq1_input /sys/class/drm/card*/device/hwmon/hwmon*/power1_average /sys/class/drm/card*/device/hwmon/hwmon*/power1_cap; do echo "$f: $(cat $f 2>&1)"; done
rocm-smi --showclocks --showpower 2>&1 | head -30
echo "== persistent chain prototype"
./prop_chain_test 207 4352 4 30
./prop_chain_test 207 4352 8 30
./prop_chain_test 207 4352 12 30
./prop_chain_test 207 8448 4 30
./prop_chain_test 207 8448 12 30
echo "== clock under the K loop: base / no DMA / no MFMA (X3 cfg 3, N 2048, 3 segments; plain cfg 4)"
X3=1 ./bf16_abl8 7372 2048 1843 3 nn 3 1 20 | tail -2
X3=1 ./bf16_abl9 7372 2048 1843 3 nn 3 1 20 | tail -2
X3=1 ./bf16_abl12 7372 2048 1843 3 nn 3 1 20 | tail -2
./bf16_abl8 7372 2048 1843 3 nn 4 1 20 | tail -2
./bf16_abl9 7372 2048 1843 3 nn 4 1 20 | tail -2
./bf16_abl12 7372 2048 1843 3 nn 4 1 20 | tail -2
echo "== G0 issues its DMA before its fragment reads (A/B), X3 cfg 3 / 7, plain cfg 4 / 3"
for b in bf16_abl0 bf16_ablg0; do
  echo "-- $b"
  X3=1 ./$b 7372 2048 1843 1 nn 3 1 20 | tail -1
  X3=1 ./$b 7372 2048 1843 3 nn 3 1 20 | tail -1
  X3=1 ./$b 7372 1024 1843 1 nn 7 1 20 | tail -1
  X3=1 ./$b 1843 2048 1843 4 nn 3 2 20 | tail -1
  ./$b 7372 2048 1843 1 nn 4 1 20 | tail -1
  ./$b 7372 2048 1843 1 nn 3 1 20 | tail -1
  ./$b 7372 1024 1843 1 nn 7 1 20 | tail -1
done
X3=1 ./bf16_ablg08 7372 2048 1843 3 nn 3 1 20 | tail -2
} > $out/r6c.log 2>&1
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > $out/r6c_tests.log 2>&1
tail -15 $out/r6c_tests.log
cat $out/r6c.log
