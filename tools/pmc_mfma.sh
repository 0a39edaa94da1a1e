#!/bin/bash
# MFMA utilisation of the IN-MODEL launches of a short bench run (north_star: "rocprof ... MFMA utilisation"):
#   SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES per kernel, over the LAST `keep` dispatches of every kernel name (the train
#   steps at the end of the run; the one-off tile autotune at its start launches the same kernels on scratch data).
# Counters only (--pmc with --kernel-trace; no other trace domain).  usage: tools/pmc_mfma.sh <tag> [bench args]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmcm_${tag} -o r -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary --no-regimes "$@" > /dev/null 2>&1
python3 - "$tag" <<'PY'
import csv, glob, sys, os, collections, re
tag = sys.argv[1]
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out"
rows = collections.defaultdict(lambda: collections.defaultdict(dict))      # name -> dispatch -> counter -> value
for f in glob.glob(f"{root}/pmcm_{tag}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void mcrn::", "").replace("mcrn::", "")
        rows[n][int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
out = []
for n, disp in rows.items():
    ids = sorted(disp)
    keep = ids[-min(len(ids), 48):]
    agg = collections.Counter()
    for i in keep:
        for c, v in disp[i].items():
            agg[c] += v
    k = len(keep)
    busy, mf = agg.get("SQ_BUSY_CYCLES", 0.0), agg.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    if mf <= 0:
        continue
    out.append((mf, n, k, mf / k, busy / k, agg.get("GRBM_GUI_ACTIVE", 0.0) / k, agg.get("SQ_WAVE_CYCLES", 0.0) / k,
                agg.get("SQ_WAIT_ANY", 0.0) / k, agg.get("SQ_WAIT_INST_ANY", 0.0) / k))
lines = ["MFMA utilisation of the in-model launches (last <= 48 dispatches of each kernel; counters are chip totals per dispatch):",
         "  util = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES / 32 shader engines x 1024 SIMDs)   [SQ_BUSY_CYCLES sums the busy cycles of the 32 SEs:",
         "  SQ_BUSY_CYCLES / 32 is the kernel's duration in shader cycles; MFMA_BUSY counts 32 cycles per v_mfma_f32_32x32x16_bf16 per SIMD]",
         "kernel | dispatches averaged | SQ_VALU_MFMA_BUSY_CYCLES | SQ_BUSY_CYCLES | duration (cycles) | MFMA util | WAIT_ANY/WAVE_CYCLES | WAIT_INST_ANY/WAVE_CYCLES"]
for mf, n, k, mfa, busy, gui, wc, wa, wi in sorted(out, reverse=True):
    if k < 40 and "gemm_bf16" in n:
        continue                                   # tile configurations only the autotuner launched
    dur = busy / 32.0
    util = mfa / (dur * 1024.0) if dur else 0.0
    lines.append(f"{n[:72]:72s} | {k:3d} | {mfa:14.0f} | {busy:12.0f} | {dur:9.0f} | {100 * util:5.1f} % | {wa / wc if wc else 0:5.2f} | {wi / wc if wc else 0:5.2f}")
    if len(lines) > 80:
        break
open(f"{root}/mfma_{tag}.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmcm_${tag}
