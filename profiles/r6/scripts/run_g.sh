#!/bin/bash
# round 6, call G: slots 10 / 11 (three-stage 128 x 256 and 256 x 128 ping-pong tiles) against the other 32K-output tiles on the encoder products;
# the full GPU suite on the library with G1L and the new slots; EXPY-TKY in both arithmetics
out=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT/tools/kbench
{
echo "== plain"
for cfg in 1 7 9 10 11 13 14; do
  ./bf16_new2 7372 1024 1843 1 nn $cfg 1 20 | tail -2
  CB_ONLY=1 ./bf16_new2 7372 1024 1843 1 nn $cfg 1 20 1 | tail -1
  ./bf16_new2 1843 1024 1843 4 nn $cfg 4 20 | tail -1
  ./bf16_new2 7372 1843 1024 12 nt $cfg 1 5 | tail -1
done
echo "== X3"
for cfg in 1 7 9 10 11 13 14; do
  X3=1 ./bf16_new2 7372 1024 1843 1 nn $cfg 1 20 | tail -2
  X3=1 ./bf16_new2 1843 1024 1843 4 nn $cfg 4 20 | tail -1
  X3=1 ./bf16_new2 7372 1843 1024 12 nt $cfg 1 5 | tail -1
done
echo "== ragged"
./bf16_new2 333 136 77 2 nn 10 1 5 1
./bf16_new2 700 333 200 2 nt 11 3 5
X3=1 ./bf16_new2 333 136 77 2 nn 10 1 5 1
X3=1 ./bf16_new2 700 333 200 2 nt 11 3 5
X3=1 ./bf16_new2 300 200 88 3 nt 10 2 5
} > $out/r6g.log 2>&1
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > $out/r6g_tests.log 2>&1
tail -5 $out/r6g_tests.log
MCRN_TUNE_LOG=1 python bench.py --config expytky --precision bf16x3 --no-cpu-baseline > $out/r6g_expytky_x3.json 2> $out/r6g_expytky_x3.err
MCRN_TUNE_LOG=1 python bench.py --config expytky --no-cpu-baseline > $out/r6g_expytky.json 2> $out/r6g_expytky.err
python - <<'PY'
import json, os
o = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/"
for f in ("r6g_expytky_x3", "r6g_expytky"):
    d = json.load(open(o + f + ".json")); r = d["roofline"]
    print(f, d["value"], d["ms_per_step"], r["frac"], r.get("shader_clock_mhz"), r.get("frac_of_mfma_peak_at_held_clock"), r["avg_launch_us"], d.get("value_no_teacher"), d.get("eval_samples_per_s"))
PY
cat $out/r6g.log | grep -v "^M="
