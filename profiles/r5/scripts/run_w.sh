#!/bin/bash
# round 5, call W: bf16 mode, width of the decoder's deferred weight gradients - finer sweep
out=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
q() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d.get('value_no_teacher'), d.get('kernel_launches_per_step'))"; }
{
for rep in 1 2 3; do
for w in 256 192 128 96 72; do
MCRN_DEC_WG_BF16=$w python bench.py --config expytky --no-secondary --no-cpu-baseline --no-roofline --no-syn --no-regimes 2>/dev/null | q "expytky dec_wg=$w"
done
done
} > $out/r5w.log 2>&1
cat $out/r5w.log
