#!/bin/bash
# one GPU-box pass that produces everything profiles/rN/ holds: bench lines of the four BASELINE configs, rocprofv3 kernel
# stats / steady-state tables / gap analysis, PMC HBM traffic, in-model MFMA utilisation.  usage: tools/measure_round.sh <tag>
tag=${1:-rX}
out=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
python bench.py > $out/${tag}_bench_metrla.json 2> $out/${tag}_bench_metrla.err
python bench.py --config pemsbay > $out/${tag}_bench_pemsbay.json 2> /dev/null
python bench.py --config expytky > $out/${tag}_bench_expytky.json 2> /dev/null
python bench.py --config syn8192 --steps 5 --warmup 2 > $out/${tag}_bench_syn8192.json 2> /dev/null
for cfg in metrla pemsbay expytky; do bash tools/prof_stats.sh ${tag}_$cfg --config $cfg > /dev/null 2>&1; done
WINDOW_MS=900 bash tools/prof_stats.sh ${tag}_syn8192 --config syn8192 > /dev/null 2>&1
for cfg in metrla pemsbay expytky; do bash tools/pmc_traffic.sh ${tag}_$cfg --config $cfg > $out/${tag}_traffic_$cfg.log 2>&1; done
bash tools/pmc_mfma.sh ${tag}_expytky --config expytky > /dev/null 2>&1
# the regime training mostly runs (no step teacher-forced: model/MegaCRN.py:146-147 beyond ~30 000 batches): steady kernel table
bash tools/prof_stats.sh ${tag}_expytky_noteacher --config expytky --batches-seen 1000000 --no-regimes > /dev/null 2>&1
bash tools/prof_stats.sh ${tag}_metrla_noteacher --config metrla --batches-seen 1000000 --no-regimes --no-secondary > /dev/null 2>&1
bash tools/pmc_mfma.sh ${tag}_metrla --config metrla > /dev/null 2>&1
ls $out | grep "^${tag}_\|_${tag}_" | head -60
