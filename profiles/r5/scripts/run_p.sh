#!/bin/bash
# round 5, call P: tail of the small-graph step (encoder gate weight gradient on the second helper queue) A/B; what the host does at the
# step boundary; the bf16 tuner's candidate table at EXPY-TKY
out=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
q() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d.get('value_no_teacher'))"; }
{
echo "== parity"
timeout 900 python -m pytest tests -m gpu -x -q -k "model_train_step or golden or trajectory or full_size_metrla or half_batches" 2>&1 | tail -3
echo "== A/B tail"
for rep in 1 2 3; do
for c in metrla pemsbay; do
python bench.py --config $c --no-secondary --no-cpu-baseline --no-roofline --no-syn 2>/dev/null | q "$c new"
MEGACRN_LIB=$GRAFT_REPO_ROOT/megacrn_amd/libmegacrn_hip_prev.so python bench.py --config $c --no-secondary --no-cpu-baseline --no-roofline --no-syn 2>/dev/null | q "$c prev"
done
done
echo "== step boundary"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --hip-trace --kernel-trace --memory-copy-trace --output-format csv -d /tmp/sb -o r -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-secondary --no-cpu-baseline --no-roofline --no-regimes --no-syn > /dev/null 2>&1
python3 $GRAFT_REPO_ROOT/tools/r5/step_boundary.py /tmp/sb
cd $GRAFT_REPO_ROOT
python tools/timeline.py --help > /dev/null 2>&1
echo "== tuner table, EXPY-TKY bf16"
MCRN_TUNE_LOG=1 python bench.py --config expytky --no-secondary --no-cpu-baseline --no-roofline --no-regimes --no-syn 2>&1 >/dev/null | grep "mcrn tune" | head -60
} > $out/r5p.log 2>&1
tail -150 $out/r5p.log
