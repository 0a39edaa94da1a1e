#!/bin/bash
# round 6, call D: 16-deep hi/lo K tiles in four stages + group 0's refill between its MFMAs (G0C) against the two-stage 32-deep forms and against
# the same code with G0C off; the rest of the GPU suite (call C stopped at a wrong expectation of a new assertion); bench lines
out=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT/tools/kbench
{
for b in bf16_new bf16_new_g0c0; do
echo "== $b: X3"
for cfg in 3 4 5 6 7 8 9; do
  X3=1 ./$b 7372 1024 1843 1 nn $cfg 1 20 | tail -2
  X3=1 ./$b 7372 2048 1843 1 nn $cfg 1 20 | tail -1
  X3=1 ./$b 1843 1024 1843 4 nn $cfg 4 20 | tail -1
  X3=1 ./$b 1843 2048 1843 4 nn $cfg 4 20 | tail -1
  X3=1 ./$b 7372 1843 1024 12 nt $cfg 1 5 | tail -1
done
echo "== $b: plain, 4-stage forms (cfg 3, 5, 6, 7) and the 2-stage ones (4, 8, 9)"
for cfg in 3 4 5 6 7 8 9; do
  ./$b 7372 1024 1843 1 nn $cfg 1 20 | tail -1
  ./$b 7372 2048 1843 1 nn $cfg 1 20 | tail -1
  ./$b 1843 2048 1843 4 nn $cfg 2 20 | tail -1
  ./$b 32768 4096 8192 1 nn $cfg 1 5 | tail -1
done
done
echo "== ragged, new forms"
X3=1 ./bf16_new 300 200 88 3 nt 3 2 5
X3=1 ./bf16_new 333 136 77 2 nn 9 1 5 1
X3=1 ./bf16_new 700 333 200 2 nt 5 3 5
X3=1 ./bf16_new 1000 520 40 1 nn 6 1 5 1
X3=1 ./bf16_new 1000 520 8 1 nn 3 1 5 1
./bf16_new 333 136 77 2 nn 3 1 5 1
./bf16_new 700 333 200 2 nt 7 3 5
} > $out/r6d.log 2>&1
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -x -q -k "bf16_mode or half_batches or harness or attached or alternative or variants" > $out/r6d_tests.log 2>&1
tail -8 $out/r6d_tests.log
python bench.py --no-secondary --no-syn --no-cpu-baseline > $out/r6d_metrla.json 2> /dev/null
MCRN_TUNE_LOG=1 python bench.py --config expytky --precision bf16x3 --no-cpu-baseline > $out/r6d_expytky_x3.json 2> $out/r6d_expytky_x3.err
python bench.py --config expytky --no-cpu-baseline > $out/r6d_expytky.json 2> /dev/null
python - <<'PY'
import json, os
o = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/"
for f in ("r6d_metrla", "r6d_expytky_x3", "r6d_expytky"):
    d = json.load(open(o + f + ".json")); r = d["roofline"]
    print(f, d["value"], d["ms_per_step"], r["frac"], r.get("shader_clock_mhz"), r.get("frac_of_mfma_peak_at_held_clock"), r["avg_launch_us"], d.get("value_no_teacher"), d.get("eval_samples_per_s"))
PY
grep -c OK $out/r6d.log; grep -B1 BAD $out/r6d.log
