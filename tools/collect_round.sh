#!/bin/bash
# Copy the products of tools/measure_round.sh <tag> from gpurun_out/ into profiles/<round>/ under the names its README lists.
# Nothing is rewritten on the way (the bench lines are what bench.py printed); a missing product is an error.
# usage: tools/collect_round.sh <tag> <round dir>      e.g.  tools/collect_round.sh r5 profiles/r5
set -euo pipefail
tag=${1:?tag}; dst=${2:?round dir}; src=gpurun_out
mkdir -p $dst
cpy() { [ -s "$1" ] || { echo "collect_round: missing $1" >&2; exit 1; }; cp "$1" "$2"; }
for c in metrla pemsbay expytky syn8192; do
  [ -s $src/${tag}_bench_$c.json ] || { echo "collect_round: missing $src/${tag}_bench_$c.json" >&2; exit 1; }
  tail -1 $src/${tag}_bench_$c.json > $dst/bench_$c.json
  cpy $src/${tag}_${c}_kernel_stats.csv $dst/${c}_kernel_stats.csv
  cpy $src/${tag}_${c}_steady.txt $dst/${c}_steady_kernels.txt
  cpy $src/${tag}_${c}_timeline.txt $dst/${c}_timeline.txt
  cpy $src/${tag}_${c}_gaps.txt $dst/${c}_gaps.txt
done
for c in metrla pemsbay expytky expytky_bf16x3; do cpy $src/traffic_${tag}_$c.json $dst/traffic_$c.json; done
cpy $src/${tag}_expytky_bf16x3_steady.txt $dst/expytky_bf16x3_steady_kernels.txt
for c in metrla expytky; do cpy $src/mfma_${tag}_$c.txt $dst/mfma_inmodel_$c.txt; done
mkdir -p profiles/tiles; cpy $src/${tag}_tiles_syn8192_B32_bf16.json profiles/tiles/syn8192_B32_bf16.json
[ -s $src/${tag}_default_command_kernel_stats.csv ] && cp $src/${tag}_default_command_kernel_stats.csv $dst/default_command_kernel_stats.csv
[ -s $src/${tag}_metrla_noteacher_steady.txt ] && cp $src/${tag}_metrla_noteacher_steady.txt $dst/metrla_noteacher_steady_kernels.txt
ls $dst
