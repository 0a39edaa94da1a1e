#!/bin/bash
# One GPU-box pass that produces everything profiles/<round>/ holds.  Order matters: the PMC traffic tables are collected FIRST and
# copied into profiles/<round>/ of the box's snapshot, so that the bench lines that follow read them themselves (roofline.traffic,
# step_fabric_gb) - no line is rewritten after it was printed.  Fails loudly on a missing product.
# usage: MCRN_GIT_REV=$(git rev-parse --short HEAD) tools/measure_round.sh <tag> <round dir>        e.g.  tools/measure_round.sh r6 profiles/r6
set -euo pipefail
tag=${1:?tag}; dst=${2:?round dir}
out=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
mkdir -p $dst
need() { [ -s "$1" ] || { echo "measure_round: missing product $1" >&2; exit 1; }; }
# 0. GEMM tile table of the N = 8192 shape, written by THIS build (bench.py adopts a table only when its build id is the library's): with it
#    the default line carries the syn8192 leg without a minute of tuning
python bench.py --config syn8192 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-regimes --save-tiles profiles/tiles/syn8192_B32_bf16.json > /dev/null 2> $out/${tag}_tiles.err
need profiles/tiles/syn8192_B32_bf16.json; cp profiles/tiles/syn8192_B32_bf16.json $out/${tag}_tiles_syn8192_B32_bf16.json
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
# 3b. the driver's own command under rocprofv3 (kernel stats of exactly what BENCH_rNN.json times)
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_${tag}_default -o r -- python3 $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 > $out/${tag}_default_command.json 2> /dev/null )
f=$(find $out/prof_${tag}_default -name "*kernel_stats.csv" | head -1); cp "$f" $out/${tag}_default_command_kernel_stats.csv; rm -rf $out/prof_${tag}_default
# 4. in-model MFMA utilisation
bash tools/pmc_mfma.sh ${tag}_expytky --config expytky > /dev/null 2>&1; need $out/mfma_${tag}_expytky.txt
bash tools/pmc_mfma.sh ${tag}_metrla --config metrla > /dev/null 2>&1; need $out/mfma_${tag}_metrla.txt
ls $out | grep "^${tag}_\|_${tag}_" | head -80
