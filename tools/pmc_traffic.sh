#!/bin/bash
# HBM traffic of every kernel of a short bench run: FETCH_SIZE and WRITE_SIZE in separate passes
# (MI355X_MICROARCH.md: TCC has 4 slots, FETCH_SIZE costs 3, WRITE_SIZE 2).  usage: tools/pmc_traffic.sh <tag> [bench args]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
i=1
for P in "FETCH_SIZE" "WRITE_SIZE"; do
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmct_${tag}_$i -o r -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-regimes "$@" > /dev/null 2>&1
  i=$((i+1))
done
python3 - "$tag" <<'PY'
import csv, glob, sys, os, collections, re, json
tag = sys.argv[1]
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out"
out = collections.defaultdict(dict); cnt = collections.Counter()
for i in (1, 2):
    for f in glob.glob(f"{root}/pmct_{tag}_{i}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void mcrn::", "").replace("mcrn::", "")
            acc[(n, r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (n, c), v in acc.items():
            out[n][c] = sum(v) / len(v); cnt[n] = len(v)
res = {}
for n, d in out.items():
    f, w = d.get("FETCH_SIZE", 0.0), d.get("WRITE_SIZE", 0.0)
    # units: KiB.  gfx950 FETCH_SIZE counts 64 B per 128-B request on wide coalesced reads -> x2 (guide, HBM section)
    res[n] = {"launches": cnt[n], "FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w,
              "hbm_bytes_per_launch_corrected": (2 * f + w) * 1024}
json.dump(res, open(f"{root}/traffic_{tag}.json", "w"), indent=1, sort_keys=True)
for n, d in sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch_corrected"] * kv[1]["launches"])[:12]:
    print(f"{n[:60]:60s} n={d['launches']:4d} fetch={d['FETCH_SIZE_KiB']/1024:8.2f}MiB write={d['WRITE_SIZE_KiB']/1024:8.2f}MiB")
PY
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmct_${tag}_[12]
