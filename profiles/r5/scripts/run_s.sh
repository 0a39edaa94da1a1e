#!/bin/bash
# round 5, call S: the round's measurement pass (tools/measure_round.sh), final library
out=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
bash tools/measure_round.sh r5 profiles/r5 > $out/r5s_measure.log 2>&1
echo "measure_round rc=$?" >> $out/r5s_measure.log
tail -30 $out/r5s_measure.log
