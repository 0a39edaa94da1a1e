#!/bin/bash
# Fabric-side (HBM + Infinity Cache) traffic of ONE steady-state train step, per kernel and in total.
#   FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes (MI355X_MICROARCH.md: TCC has 4 slots, FETCH_SIZE costs 3,
#   WRITE_SIZE 2; FETCH_SIZE x2 on gfx950 for wide coalesced reads), counters with --kernel-trace only.
#   Only the dispatches of the LAST train step of the run are kept: the window between the last two k_clip_adam dispatches
#   (one per optimizer step).  The one-off tile autotune, the warm-up steps and the other legs never enter the table.
# usage: tools/pmc_traffic.sh <tag> [bench args]      ->  gpurun_out/traffic_<tag>.json  (+ "_step": the sum over the step)
tag=$1; shift
set -e
cd /tmp && export TMPDIR=/tmp
i=1
for P in "FETCH_SIZE" "WRITE_SIZE"; do
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmct_${tag}_$i -o r -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-regimes --no-secondary --no-syn "$@" > /dev/null 2>&1
  i=$((i+1))
done
python3 - "$tag" <<'PY'
import csv, glob, sys, os, collections, re, json
tag = sys.argv[1]
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out"


def norm(n):
    """The two passes are two runs with two tile autotunes: a tiled GEMM of the same role may run a different tile configuration in each.
    Tile parameters are folded away; what identifies a launch is the kernel family, its operand forms and its ROLE."""
    m = re.match(r"^(gemm_f32_kernel|gemm_bf16x3_kernel)<.*, (true|false), (true|false), (\d+)>$", n)
    if m:
        return f"{m.group(1)}<*, {m.group(2)}, {m.group(3)}, {m.group(4)}>"
    m = re.match(r"^gemm_bf16(?:_pp)?_kernel<.*, (true|false), (\d+)(?:, (true|false))?>$", n)      # <tile..., BTR, ROLE[, X3]>
    if m:
        return f"gemm_bf16*<*, {m.group(1)}, {m.group(2)}>" + (" hi/lo" if m.group(3) == "true" else "")
    return n


def step_window(counter):
    """[(kernel name, value)] of the last train step of one pass, in dispatch order"""
    rows = []
    for f in glob.glob(f"{root}/pmct_{tag}_{1 if counter == 'FETCH_SIZE' else 2}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            n = norm(re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void mcrn::", "").replace("mcrn::", ""))
            rows.append((int(r["Dispatch_Id"]), n, float(r["Counter_Value"])))
    if not rows:
        sys.exit(f"pmc_traffic: no {counter} rows collected for {tag}")
    rows.sort()
    marks = [i for i, r in enumerate(rows) if r[1].startswith("k_clip_adam")]
    if len(marks) < 2:
        sys.exit(f"pmc_traffic: fewer than two optimizer steps in the {counter} pass of {tag}")
    return [(n, v) for _, n, v in rows[marks[-2] + 1: marks[-1] + 1]]


fetch, write = step_window("FETCH_SIZE"), step_window("WRITE_SIZE")
# (the two passes are two runs: the same launches, but dispatch ids of the helper queues interleave differently - match by kernel name)
acc = collections.defaultdict(lambda: [0, 0.0, 0, 0.0])
for n, f in fetch:
    a = acc[n]; a[0] += 1; a[1] += f
for n, w in write:
    a = acc[n]; a[2] += 1; a[3] += w
bad = {n: (a[0], a[2]) for n, a in acc.items() if a[0] != a[2]}
if bad:
    sys.exit(f"pmc_traffic: the two passes of {tag} did not launch the same step: {bad}")
acc = {n: [a[0], a[1], a[3]] for n, a in acc.items()}
res, total = {}, 0.0
for n, (k, f, w) in acc.items():
    # units: KiB.  gfx950 FETCH_SIZE counts 64 B per 128-B request on wide coalesced reads -> x2 (guide, HBM section)
    b = (2 * f + w) * 1024 / k
    res[n] = {"launches": k, "FETCH_SIZE_KiB": f / k, "WRITE_SIZE_KiB": w / k, "hbm_bytes_per_launch_corrected": b}
    total += b * k
res["_step"] = {"fabric_bytes": total, "dispatches": len(fetch), "window": "between the last two k_clip_adam dispatches"}
# which library these bytes belong to: bench.py reports them only when the library it loaded has the same build id
import ctypes
lib = ctypes.CDLL(os.environ["GRAFT_REPO_ROOT"] + "/megacrn_amd/libmegacrn_hip.so")
lib.mcrn_build_id.restype = ctypes.c_char_p
# (the GPU box has no .git: the caller passes the revision the snapshot was taken at, MCRN_GIT_REV=$(git rev-parse --short HEAD))
res["_meta"] = {"build_id": lib.mcrn_build_id().decode(), "lib_version": int(lib.mcrn_version()), "git": os.environ.get("MCRN_GIT_REV")}
json.dump(res, open(f"{root}/traffic_{tag}.json", "w"), indent=1, sort_keys=True)
print(f"{tag}: one step = {len(fetch)} dispatches, {total / 1e9:.2f} GB through the fabric ports")
for n, d in sorted(((n, d) for n, d in res.items() if n[:1] != "_"), key=lambda kv: -kv[1]["hbm_bytes_per_launch_corrected"] * kv[1]["launches"])[:14]:
    print(f"{n[:60]:60s} n={d['launches']:4d} fetch={d['FETCH_SIZE_KiB']/1024:8.2f}MiB write={d['WRITE_SIZE_KiB']/1024:8.2f}MiB  step share {d['hbm_bytes_per_launch_corrected'] * d['launches'] / total:5.1%}")
PY
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmct_${tag}_[12]
