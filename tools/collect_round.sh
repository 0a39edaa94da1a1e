#!/bin/bash
# copy the outputs of tools/measure_round.sh <tag> from gpurun_out/ into profiles/<round>/ under the names its README lists.
# usage: tools/collect_round.sh <tag> <round dir>      e.g.  tools/collect_round.sh r4 profiles/r4
tag=$1; dst=$2; src=gpurun_out
mkdir -p $dst
for c in metrla pemsbay expytky syn8192; do
  [ -f $src/${tag}_bench_$c.json ] && tail -1 $src/${tag}_bench_$c.json > $dst/bench_$c.json
  [ -f $src/${tag}_${c}_kernel_stats.csv ] && cp $src/${tag}_${c}_kernel_stats.csv $dst/${c}_kernel_stats.csv
  [ -f $src/${tag}_${c}_steady.txt ] && cp $src/${tag}_${c}_steady.txt $dst/${c}_steady_kernels.txt
  [ -f $src/${tag}_${c}_timeline.txt ] && cp $src/${tag}_${c}_timeline.txt $dst/${c}_timeline.txt
  [ -f $src/${tag}_${c}_gaps.txt ] && cp $src/${tag}_${c}_gaps.txt $dst/${c}_gaps.txt
  [ -f $src/traffic_${tag}_$c.json ] && cp $src/traffic_${tag}_$c.json $dst/traffic_$c.json
  [ -f $src/mfma_${tag}_$c.txt ] && cp $src/mfma_${tag}_$c.txt $dst/mfma_inmodel_$c.txt
  [ -f $src/${tag}_${c}_noteacher_steady.txt ] && cp $src/${tag}_${c}_noteacher_steady.txt $dst/${c}_noteacher_steady_kernels.txt
  [ -f $src/${tag}_${c}_noteacher_timeline.txt ] && cp $src/${tag}_${c}_noteacher_timeline.txt $dst/${c}_noteacher_timeline.txt
done
ls $dst
# the bench lines were produced before this round's traffic files existed: re-read roofline.traffic from the files just collected
python3 - "$dst" <<'PY'
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(sys.argv[1])), ".."))
sys.path.insert(0, os.getcwd())
import bench
dst = sys.argv[1]
for c, cfg in bench.CONFIGS.items():
    f = os.path.join(dst, f"bench_{c}.json")
    if not os.path.exists(f):
        continue
    d = json.loads(open(f).read().strip().split("\n")[-1])
    if "roofline" in d:
        t = bench.pmc_traffic(c, d["dtype"], cfg["N"])
        if t:
            d["roofline"]["traffic"] = t
    open(f, "w").write(json.dumps(d) + "\n")
PY
