#!/bin/bash
# One GPU-box pass that produces everything profiles/<round>/ holds.  Order matters: the PMC traffic tables are collected FIRST and
# copied into profiles/<round>/ of the box's snapshot, so that the bench lines that follow read them themselves (roofline.traffic,
# step_fabric_gb) - no line is rewritten after it was printed.  Fails loudly on a missing product.
# usage: tools/measure_round.sh <tag> <round dir>        e.g.  tools/measure_round.sh r5 profiles/r5
set -euo pipefail
tag=${1:?tag}; dst=${2:?round dir}
out=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
mkdir -p $dst
need() { [ -s "$1" ] || { echo "measure_round: missing product $1" >&2; exit 1; }; }
# 1. fabric traffic of one steady-state step, per config and arithmetic (tools/pmc_traffic.sh)
for cfg in metrla pemsbay expytky; do
  bash tools/pmc_traffic.sh ${tag}_$cfg --config $cfg > $out/${tag}_traffic_$cfg.log 2>&1
  need $out/traffic_${tag}_$cfg.json; cp $out/traffic_${tag}_$cfg.json $dst/traffic_$cfg.json
done
bash tools/pmc_traffic.sh ${tag}_expytky_bf16x3 --config expytky --precision bf16x3 > $out/${tag}_traffic_expytky_bf16x3.log 2>&1
need $out/traffic_${tag}_expytky_bf16x3.json; cp $out/traffic_${tag}_expytky_bf16x3.json $dst/traffic_expytky_bf16x3.json
# 2. bench lines (they read the tables just collected)
python bench.py > $out/${tag}_bench_metrla.json 2> $out/${tag}_bench_metrla.err; need $out/${tag}_bench_metrla.json
python bench.py --config pemsbay > $out/${tag}_bench_pemsbay.json 2> /dev/null; need $out/${tag}_bench_pemsbay.json
python bench.py --config expytky > $out/${tag}_bench_expytky.json 2> /dev/null; need $out/${tag}_bench_expytky.json
python bench.py --config syn8192 --steps 5 --warmup 2 > $out/${tag}_bench_syn8192.json 2> /dev/null; need $out/${tag}_bench_syn8192.json
# 3. rocprofv3 kernel stats / steady tables / timelines / gaps
for cfg in metrla pemsbay expytky; do bash tools/prof_stats.sh ${tag}_$cfg --config $cfg --no-secondary --no-syn > /dev/null 2>&1; need $out/${tag}_${cfg}_steady.txt; done
WINDOW_MS=900 bash tools/prof_stats.sh ${tag}_syn8192 --config syn8192 > /dev/null 2>&1; need $out/${tag}_syn8192_steady.txt
bash tools/prof_stats.sh ${tag}_expytky_bf16x3 --config expytky --precision bf16x3 > /dev/null 2>&1; need $out/${tag}_expytky_bf16x3_steady.txt
bash tools/prof_stats.sh ${tag}_metrla_noteacher --config metrla --batches-seen 1000000 --no-regimes --no-secondary --no-syn > /dev/null 2>&1
# 4. in-model MFMA utilisation
bash tools/pmc_mfma.sh ${tag}_expytky --config expytky > /dev/null 2>&1; need $out/mfma_${tag}_expytky.txt
bash tools/pmc_mfma.sh ${tag}_metrla --config metrla > /dev/null 2>&1; need $out/mfma_${tag}_metrla.txt
ls $out | grep "^${tag}_\|_${tag}_" | head -80
