#!/usr/bin/env python3
"""Regenerate the measured tables of README.md, DESIGN.md (section 8) and profiles/r4/README.md from profiles/r4/bench_*.json.

Every generated block sits between the markers  <!-- r4:NAME:begin -->  and  <!-- r4:NAME:end -->  of its document; the
prose around the blocks is written by hand.  Usage: python tools/fill_docs.py"""
import json
import os
import re

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
R = os.path.join(ROOT, "profiles", "r4")
L = lambda c: json.load(open(os.path.join(R, f"bench_{c}.json")))
m, p, e, s = L("metrla"), L("pemsbay"), L("expytky"), L("syn8192")
R2 = {"metrla": "10539 / 6.07 ms", "pemsbay": "5677 / 11.27 ms", "expytky": "4672 / 6.85 ms", "syn8192": "126 / 254 ms"}     # round 3
NAMES = [("metrla", "METR-LA N=207 B=64 T=12", m), ("pemsbay", "PEMS-BAY N=325 B=64 T=12", p),
         ("expytky", "EXPY-TKY N=1843 B=32 T=6 H=32", e), ("syn8192", "SYN N=8192 B=32 T=12", s)]


def roof(d):
    r = d["roofline"]
    if r["bound"] == "hbm":
        return f"HBM-bound (AI {r['arithmetic_intensity']:.0f} < 312): {r['achieved']:.0f} GB/s = {100 * r['frac']:.1f} % of 8 TB/s"
    return f"MFMA-bound (AI {r['arithmetic_intensity']:.0f}): {r['achieved']:.0f} TF = **{100 * r['frac']:.1f} %** of 2.5 PF"


def traffic(d):
    r = d["roofline"]
    if not r.get("traffic"):
        return "—"
    return f"{r['traffic'] / 1e6:.1f} MB vs {r['alg_bytes_per_launch'] / 1e6:.1f} MB algorithmic ({r['traffic'] / r['alg_bytes_per_launch']:.2f}×)"


def table(with_traffic):
    head = "| config | arithmetic | samples/s | no teacher forcing | eval forward (samples/s) | ms/step | launches/step | round 3 (samples/s / step) | forward propagation (dominant kernel), dispatch-attached HIP events in real steps |"
    sep = "|---|---|---|---|---|---|---|---|---|"
    if with_traffic:
        head += " PMC HBM bytes per launch |"
        sep += "---|"
    rows = [head, sep]
    for key, lbl, d in NAMES:
        nt, ev = d.get("value_no_teacher"), d.get("eval_samples_per_s")
        row = (f"| {lbl} | {d['dtype']} | **{d['value']:.0f}** | {'%.0f' % nt if nt else '—'} | {'%.0f' % ev if ev else '—'} | {d['ms_per_step']:.2f} | "
               f"{d['kernel_launches_per_step']} | {R2[key]} | {roof(d)} |")
        if with_traffic:
            row += f" {traffic(d)} |"
        rows.append(row)
    return "\n".join(rows)


def roles():
    out = []
    for key, lbl, d in NAMES:
        out.append(f"* {lbl.split(' ')[0]}: " + ", ".join(
            f"{k} {v['ms_per_step']:.2f} ms ({v['launches_per_step']} × {v['avg_us']:.0f} µs, {v['alg_tflops']:.0f} TF)" for k, v in d["gemm_roles"].items()))
    return "\n".join(out)


def cpu():
    out = []
    for key, lbl, d in NAMES:
        c = d.get("cpu_baseline")
        if not c:
            continue
        out.append(f"* {lbl.split(' ')[0]}: torch-CPU restatement {c['value']:.3g} samples/s on {c['cores']} threads (best of {c.get('thread_sweep', {})}), {c.get('value_1thread', float('nan')):.3g} on one thread"
                   f" (what the reference pins); numpy port, one thread: {c.get('value_numpy_port_1thread', float('nan')):.3g}")
    return "\n".join(out)


def secondary():
    sec = m.get("secondary")
    if not sec:
        return "(no secondary leg in bench_metrla.json)"
    r = sec["roofline"]
    return (f"`secondary` object of the default bench line (EXPY-TKY shape, bf16 mode, measured in the same run): {sec['value']:.0f} samples/s "
            f"({sec.get('value_no_teacher', float('nan')):.0f} without teacher forcing, eval forward {sec.get('eval_samples_per_s', float('nan')):.0f}), "
            f"{sec['ms_per_step']:.2f} ms/step, forward propagation {r['achieved']:.0f} TF = {100 * r['frac']:.1f} % of 2.5 PF.")


BLOCKS = {
    "table": lambda: table(False),
    "table_traffic": lambda: table(True),
    "roles": roles,
    "cpu": cpu,
    "secondary": secondary,
}


def fill(path):
    t = open(path).read()
    for name, fn in BLOCKS.items():
        pat = re.compile(rf"(<!-- r4:{name}:begin -->\n).*?(<!-- r4:{name}:end -->)", re.S)
        if pat.search(t):
            t = pat.sub(lambda mo: mo.group(1) + fn() + "\n" + mo.group(2), t)
    open(path, "w").write(t)


for f in ("README.md", "DESIGN.md", os.path.join("profiles", "r4", "README.md")):
    fill(os.path.join(ROOT, f))
print("filled")
