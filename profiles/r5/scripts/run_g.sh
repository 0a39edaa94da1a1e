#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
{
for cfg in metrla pemsbay; do
  timeout 600 python tools/r5/microbatch_probe.py $cfg 2 2>&1 | grep -v Warning
done
timeout 600 python tools/r5/microbatch_probe.py metrla 4 2>&1 | grep -v Warning
timeout 600 python tools/r5/microbatch_probe.py expytky 2 2>&1 | grep -v Warning
} > $out/r5g.log 2>&1
tail -40 $out/r5g.log
