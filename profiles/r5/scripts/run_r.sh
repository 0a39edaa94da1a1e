#!/bin/bash
# round 5, call R: the round's measurement pass (tools/measure_round.sh) + the N = 8192 traffic question: fabric-port bytes of the
# propagation product with the B operand inside (B = 32: 134 MB) and outside (B = 64: 268 MB + the 134 MB S stack) the 256 MiB Infinity Cache
out=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
bash tools/measure_round.sh r5 profiles/r5 > $out/r5r_measure.log 2>&1
echo "measure_round rc=$?" >> $out/r5r_measure.log
tail -5 $out/r5r_measure.log
{
bash tools/pmc_traffic.sh r5_syn8192 --config syn8192
python bench.py --config syn8192 --batch 64 --steps 2 --warmup 1 --no-secondary --no-cpu-baseline --no-regimes --no-roofline --save-tiles profiles/tiles/syn8192_B64_bf16.json 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('syn8192 B=64', d['value'], d['ms_per_step'])"
cp profiles/tiles/syn8192_B64_bf16.json $out/
bash tools/pmc_traffic.sh r5_syn8192_B64 --config syn8192 --batch 64
} > $out/r5r_syn.log 2>&1
tail -40 $out/r5r_syn.log
