#!/bin/bash
# round 5, call T: forward/backward hoisting of the decoder's input channels at METR-LA re-measured (round 4: -0.7 % / -3.8 %; since
# round 5 its backward needs no gathered hop)
out=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
q() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d.get('value_no_teacher'), d.get('kernel_launches_per_step'))"; }
{
for rep in 1 2 3; do
python bench.py --no-secondary --no-cpu-baseline --no-roofline --no-syn 2>/dev/null | q "metrla default"
MCRN_HOIST_FWD=2 python bench.py --no-secondary --no-cpu-baseline --no-roofline --no-syn 2>/dev/null | q "metrla hoisted"
done
} > $out/r5t.log 2>&1
cat $out/r5t.log
