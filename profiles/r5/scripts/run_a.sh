#!/bin/bash
# round 5, call A: packed-fp32 control on this box + same-box A/B of the library built with / without the SLP vectoriser
out=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
q() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d.get('value_no_teacher'), d.get('eval_samples_per_s'))"; }
{
echo "== packed-fp32 control (wgrad_test_slp: v_pk_* present, see profiles/r5/experiments.md)"
cd tools/kbench
for shape in "1 80000 5 28 48 256" "1 80000 5 68 64 256" "1 80000 5 28 24 256" "1 80000 5 100 32 256"; do
  for mf in 1024 4096; do
    echo -n "slp  MFMAN=$mf $shape: "; MFMAN=$mf CONC=12 NROT=1 timeout 300 ./wgrad_test_slp $shape 2 | grep "CONC:"
  done
  echo -n "noslp MFMAN=1024 $shape: "; MFMAN=1024 CONC=12 NROT=1 timeout 300 ./wgrad_test $shape 2 | grep "CONC:"
done
cd $GRAFT_REPO_ROOT
echo "== A/B bench (default flags vs SLP on)"
for rep in 1 2; do
  python bench.py --no-secondary --no-cpu-baseline --no-roofline 2>/dev/null | q "metrla default"
  MEGACRN_LIB=$GRAFT_REPO_ROOT/megacrn_amd/libmegacrn_hip_slp.so python bench.py --no-secondary --no-cpu-baseline --no-roofline 2>/dev/null | q "metrla slp"
done
python bench.py --config pemsbay --no-secondary --no-cpu-baseline --no-roofline 2>/dev/null | q "pemsbay default"
MEGACRN_LIB=$GRAFT_REPO_ROOT/megacrn_amd/libmegacrn_hip_slp.so python bench.py --config pemsbay --no-secondary --no-cpu-baseline --no-roofline 2>/dev/null | q "pemsbay slp"
python bench.py --config expytky --no-secondary --no-cpu-baseline --no-roofline 2>/dev/null | q "expytky default"
MEGACRN_LIB=$GRAFT_REPO_ROOT/megacrn_amd/libmegacrn_hip_slp.so python bench.py --config expytky --no-secondary --no-cpu-baseline --no-roofline 2>/dev/null | q "expytky slp"
echo "== parity with the SLP library (subset) + race script"
MEGACRN_LIB=$GRAFT_REPO_ROOT/megacrn_amd/libmegacrn_hip_slp.so timeout 900 python -m pytest tests -m gpu -x -q -k "(model_train_step or golden or kernel_variants or trajectory or full_size_metrla or half_batches or bf16_mode_train or large_graph or baseline_config) and not alternative_paths" 2>&1 | tail -4
# (the race replay script of rounds 2-4, tools/scratch/dbg_race.py, ran here; removed with tools/scratch at the end of round 5)
} > $out/r5a.log 2>&1
tail -50 $out/r5a.log
