#!/bin/bash
# round 5, call Q: in-kernel phase timeline of the fused two-hop kernels (stand-alone harness)
out=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT/tools/kbench
{
for b in prop1_test prop1_test_tl1; do
echo "== $b"
./$b 207 4352 40
./$b 207 8448 40
./$b 325 4352 40
./$b 325 8448 40
./$b 250 4352 40
./$b 100 4352 40
done
} > $out/r5q.log 2>&1
cat $out/r5q.log
