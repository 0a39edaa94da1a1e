#!/bin/bash
# round 5, call U: the two queue bubbles per BPTT cell of the small graphs: `ready` event attached to the fused chain's dispatch
# (MCRN_ATTACH_READY) and one plane-set pair per cell (MCRN_FLAT_PAIRS: no guard wait); A/B inside one library + timeline
out=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
q() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d.get('value_no_teacher'), d.get('kernel_launches_per_step'))"; }
{
echo "== parity (both on)"
timeout 900 python -m pytest tests -m gpu -x -q -k "model_train_step or golden or trajectory or full_size_metrla or half_batches or kernel_variants" 2>&1 | tail -3
echo "== A/B"
for rep in 1 2 3; do
for c in metrla pemsbay; do
python bench.py --config $c --no-secondary --no-cpu-baseline --no-roofline --no-syn 2>/dev/null | q "$c both"
MCRN_ATTACH_READY=0 python bench.py --config $c --no-secondary --no-cpu-baseline --no-roofline --no-syn 2>/dev/null | q "$c flat-only"
MCRN_FLAT_PAIRS=0 python bench.py --config $c --no-secondary --no-cpu-baseline --no-roofline --no-syn 2>/dev/null | q "$c attach-only"
MCRN_ATTACH_READY=0 MCRN_FLAT_PAIRS=0 python bench.py --config $c --no-secondary --no-cpu-baseline --no-roofline --no-syn 2>/dev/null | q "$c neither"
done
done
echo "== timeline (both on)"
bash tools/prof_stats.sh r5u_metrla --config metrla --no-secondary --no-syn > /dev/null 2>&1
sed -n 330,372p $out/r5u_metrla_timeline.txt | cut -c1-100
head -8 $out/r5u_metrla_gaps.txt
} > $out/r5u.log 2>&1
tail -90 $out/r5u.log
