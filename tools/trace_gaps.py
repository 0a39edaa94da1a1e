#!/usr/bin/env python3
"""Gap analysis of a rocprofv3 kernel trace (kernel_trace.csv): GPU busy time (union over queues), idle time,
per-queue busy time and the distribution of gaps between consecutive kernels, over the last `steps` train steps.
Usage: python tools/trace_gaps.py <kernel_trace.csv> [first_fraction_to_skip=0.5]"""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")))
rows.sort()
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = rows[int(len(rows) * skip):]
t0, t1 = rows[0][0], max(r[1] for r in rows)
span = t1 - t0
busy, cur_s, cur_e = 0, rows[0][0], rows[0][1]
gaps = []
for s, e, _, _ in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append(s - cur_e)
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"kernels {len(rows)}  span {span/1e6:.2f} ms  busy(union) {busy/1e6:.2f} ms ({100*busy/span:.1f}%)  idle {sum(gaps)/1e6:.2f} ms in {len(gaps)} gaps")
gaps.sort()
if gaps:
    n = len(gaps)
    print("gap ns: p10 %d  p50 %d  p90 %d  p99 %d  max %d" % (gaps[n // 10], gaps[n // 2], gaps[9 * n // 10], gaps[min(n - 1, 99 * n // 100)], gaps[-1]))
perq = defaultdict(int)
for s, e, _, q in rows:
    perq[q] += e - s
for q, v in sorted(perq.items()):
    print(f"queue {q}: kernel time {v/1e6:.2f} ms ({100*v/span:.1f}% of span)")
pern = defaultdict(lambda: [0, 0])
for s, e, k, _ in rows:
    pern[k.split("(")[0][-70:]][0] += e - s
    pern[k.split("(")[0][-70:]][1] += 1
for k, (v, c) in sorted(pern.items(), key=lambda kv: -kv[1][0])[:22]:
    print(f"{100*v/span:5.1f}%  {v/c/1e3:8.1f} us x {c:5d}  {k}")

# per-queue view: idle time between consecutive kernels of the busiest queue, grouped by (previous -> next) kernel
import re as _re
qmain = max(perq, key=perq.get)
seq = sorted([r for r in rows if r[3] == qmain])
pair = defaultdict(lambda: [0, 0])
short = lambda k: _re.sub(r"^void ", "", k.split("(")[0]).replace("mcrn::", "")[:44]
tot_gap = 0
for a, b_ in zip(seq, seq[1:]):
    g_ = b_[0] - a[1]
    if g_ > 0:
        pair[(short(a[2]), short(b_[2]))][0] += g_
        pair[(short(a[2]), short(b_[2]))][1] += 1
        tot_gap += g_
print(f"queue {qmain}: idle between its own kernels {tot_gap/1e6:.2f} ms ({100*tot_gap/span:.1f}% of span); top (prev -> next):")
for (a, b_), (v, c) in sorted(pair.items(), key=lambda kv: -kv[1][0])[:14]:
    print(f"  {100*v/span:4.1f}%  {v/c/1e3:7.1f} us x {c:4d}  {a} -> {b_}")
