#!/bin/bash
# usage: tools/pmc_gemm.sh <tag> M N K tA tB cfg   -> gpurun_out/pmc_<tag>_{1,2,3}.csv (kernel rows only)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA"
P2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES"
P3="TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE TA_TA_BUSY_sum"
i=1
for P in "$P1" "$P2" "$P3"; do
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}_$i -o r -- python3 $GRAFT_REPO_ROOT/tools/one_gemm.py "$@" 8 > /dev/null 2>&1
  i=$((i+1))
done
python3 - "$tag" <<'PY'
import csv, glob, sys, os, collections
tag = sys.argv[1]
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out"
out = {}
for i in (1, 2, 3):
    for f in glob.glob(f"{root}/pmc_{tag}_{i}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "gemm" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            out[k] = sum(v) / len(v)
print(tag, {k: round(v) for k, v in sorted(out.items())})
PY
