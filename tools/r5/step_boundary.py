#!/usr/bin/env python3
"""Which host-side HIP calls sit between the last launch of one train step (k_clip_adam) and the first launch of the next, and how
long each takes: the step boundary showed ~180 us of GPU idle in the kernel timeline (profiles/r5/metrla_gaps.txt).
usage (GPU box):  rocprofv3 --hip-trace --kernel-trace --memory-copy-trace --output-format csv -d DIR -o r -- python3 bench.py ... ;
                  python3 tools/r5/step_boundary.py DIR"""
import csv, glob, sys, collections
d = sys.argv[1]
api = []
for f in glob.glob(f"{d}/**/*hip_api_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        api.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"], int(r["Correlation_Id"])))
api.sort()
kern = {}
for f in glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        kern[int(r["Correlation_Id"])] = (r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]))
cop = []
for f in glob.glob(f"{d}/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        cop.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", "?"), int(r["Correlation_Id"])))
print(f"{len(api)} api calls, {len(kern)} kernels, {len(cop)} copies")
# indices of the launches of k_clip_adam
marks = [i for i, a in enumerate(api) if a[3] in kern and "k_clip_adam" in kern[a[3]][0]]
print(f"{len(marks)} optimizer steps")
for m in marks[-4:-1]:
    t0 = api[m][0]
    kend = kern[api[m][3]][2]
    print(f"--- after k_clip_adam (host call at 0 us; kernel ended at +{(kend - t0) / 1e3:.1f} us)")
    n = 0
    for a in api[m + 1:]:
        name = kern[a[3]][0][:50] if a[3] in kern else ""
        extra = ""
        if a[3] in kern:
            extra = f"   kernel ran +{(kern[a[3]][1] - t0) / 1e3:.1f} .. +{(kern[a[3]][2] - t0) / 1e3:.1f}"
        for c in cop:
            if c[3] == a[3]:
                extra += f"   copy {c[2]} ran +{(c[0] - t0) / 1e3:.1f} .. +{(c[1] - t0) / 1e3:.1f}"
        print(f"  +{(a[0] - t0) / 1e3:8.1f} us  {(a[1] - a[0]) / 1e3:7.1f} us  {a[2]:32s} {name}{extra}")
        n += 1
        if n > 40 or ("gemm_f32" in name):
            break
# host enqueue time of one step vs its GPU time
if len(marks) > 3:
    a, b = marks[-3], marks[-2]
    print(f"host: one step's calls span {(api[b][0] - api[a][0]) / 1e3:.0f} us; GPU: clip_adam to clip_adam {(kern[api[b][3]][2] - kern[api[a][3]][2]) / 1e3:.0f} us")
    cnt = collections.Counter(x[2] for x in api[a:b])
    tot = collections.Counter()
    for x in api[a:b]:
        tot[x[2]] += x[1] - x[0]
    for k, v in tot.most_common(8):
        print(f"   {k:34s} n={cnt[k]:5d} total {v / 1e3:8.0f} us")
