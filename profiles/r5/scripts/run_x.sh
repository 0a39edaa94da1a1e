#!/bin/bash
# round 5, call X: full GPU suite on the final library, then the round's measurement pass
out=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
t0=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -x -q > $out/r5x_suite.log 2>&1
echo "suite rc=$? wall $(( $(date +%s) - t0 )) s" >> $out/r5x_suite.log
tail -5 $out/r5x_suite.log
bash tools/measure_round.sh r5 profiles/r5 > $out/r5s_measure.log 2>&1
echo "measure_round rc=$?" >> $out/r5s_measure.log
tail -3 $out/r5s_measure.log
