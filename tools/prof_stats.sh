#!/bin/bash
# per-kernel time of a short bench run: rocprofv3 --kernel-trace --stats.  usage (GPU box): tools/prof_stats.sh <tag> [bench args]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_${tag} -o r -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-regimes "$@" > $GRAFT_REPO_ROOT/gpurun_out/prof_${tag}_bench.json 2> /dev/null
f=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_${tag} -name "*kernel_stats.csv" | head -1)
cp "$f" $GRAFT_REPO_ROOT/gpurun_out/${tag}_kernel_stats.csv
t=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_${tag} -name "*kernel_trace.csv" | head -1)
python3 $GRAFT_REPO_ROOT/tools/trace_gaps.py "$t" 0.5 > $GRAFT_REPO_ROOT/gpurun_out/${tag}_gaps.txt 2>&1
python3 - "$t" "$GRAFT_REPO_ROOT/gpurun_out/${tag}_steady.txt" "${WINDOW_MS:-60}" <<'PY'
# per-kernel totals over the last WINDOW_MS ms of the trace (steady-state train steps: warm-up and the one-off tile autotune excluded)
import csv, sys, re, collections
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
t_mid = rows[-1][1] - int(float(sys.argv[3]) * 1e6)       # the last <window> ms of the trace
rows = [r for r in rows if r[0] >= t_mid]
acc = collections.defaultdict(lambda: [0, 0])
for s_, e_, n in rows:
    n = re.sub(r"\(.*", "", n).replace("void mcrn::", "").replace("mcrn::", "")
    acc[n][0] += 1; acc[n][1] += e_ - s_
tot = sum(v[1] for v in acc.values())
out = [f"steady-state window {(rows[-1][1] - rows[0][0]) / 1e6:.2f} ms, kernel time {tot / 1e6:.2f} ms, {len(rows)} launches"]
for n, (c, d) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:40]:
    out.append(f"{n[:84]:84s} n={c:5d} avg={d / c / 1e3:9.1f}us tot={d / 1e6:8.2f}ms {100 * d / tot:5.1f}%")
open(sys.argv[2], "w").write("\n".join(out) + "\n")
print("\n".join(out[:32]))
PY
python3 - "$t" "$GRAFT_REPO_ROOT/gpurun_out/${tag}_timeline.txt" "${TIMELINE_MS:-8}" <<'PY'
# launch-by-launch timeline of the last TIMELINE_MS ms (about one train step): start offset, duration, gap to the previous kernel
import csv, sys, re
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1])))
t0 = rows[-1][1] - int(float(sys.argv[3]) * 1e6)
rows = [r for r in rows if r[0] >= t0]
out, prev = [], rows[0][0]
for s_, e_, n in rows:
    n = re.sub(r"\(.*", "", n).replace("void mcrn::", "").replace("mcrn::", "")
    out.append(f"{(s_ - rows[0][0]) / 1e3:9.1f}us  dur {(e_ - s_) / 1e3:7.1f}us  gap {(s_ - prev) / 1e3:6.1f}us  {n[:90]}")
    prev = e_
open(sys.argv[2], "w").write("\n".join(out) + "\n")
PY
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_${tag}
