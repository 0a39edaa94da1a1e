#!/bin/bash
# PMC counters for every kernel of a short bench run, aggregated per kernel name.
# usage (GPU box): tools/pmc_bench.sh <tag> [bench args]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA"
P2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES"
P3="TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum GRBM_GUI_ACTIVE"
i=1
for P in "$P1" "$P2" "$P3"; do
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmcb_${tag}_$i -o r -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline "$@" > /dev/null 2>&1
  i=$((i+1))
done
python3 - "$tag" <<'PY'
import csv, glob, sys, os, collections, re
tag = sys.argv[1]
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out"
out = collections.defaultdict(dict); cnt = collections.Counter()
for i in (1, 2, 3):
    for f in glob.glob(f"{root}/pmcb_{tag}_{i}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void mcrn::", "").replace("mcrn::", "")
            acc[(n, r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (n, c), v in acc.items():
            out[n][c] = sum(v) / len(v); cnt[n] = len(v)
with open(f"{root}/pmcb_{tag}.txt", "w") as fo:
    for n, d in sorted(out.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0) * cnt[kv[0]]):
        line = f"{n[:58]:58s} n={cnt[n]:5d} " + " ".join(f"{k.replace('SQ_','').replace('_sum','')}={int(v)}" for k, v in sorted(d.items()))
        fo.write(line + "\n")
print(open(f"{root}/pmcb_{tag}.txt").read()[:6000])
PY
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmcb_${tag}_[123]
