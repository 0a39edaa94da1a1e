#!/bin/bash
# LDS / MFMA counters of one bf16 GEMM configuration.  usage: tools/kbench/pmc_bf16.sh <out> <harness args...>
out=$1; shift
cd /tmp && export TMPDIR=/tmp
for P in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU"; do
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d /tmp/pmcb -o r -- $GRAFT_REPO_ROOT/tools/kbench/bf16_gemm_test "$@" > /dev/null 2>&1
  python3 - <<'PY' >> $GRAFT_REPO_ROOT/gpurun_out/$out
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("/tmp/pmcb/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_bf16" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(f"{k:28s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
  rm -rf /tmp/pmcb
done
