#!/usr/bin/env python3
"""Results block of profiles/<round>/README.md, generated from the bench lines and traffic tables that tools/collect_round.sh put there
(numbers are never copied by hand).  usage: tools/round_readme.py profiles/r6 [profiles/r5]  -> markdown on stdout"""
import json
import os
import sys

CFGS = [("metrla", "METR-LA N=207 B=64 T=12"), ("pemsbay", "PEMS-BAY N=325 B=64 T=12"), ("expytky", "EXPY-TKY N=1843 B=32 T=6 H=32"),
        ("syn8192", "SYN N=8192 B=32 T=12")]


def load(d, name):
    p = os.path.join(d, f"bench_{name}.json")
    return json.loads(open(p).read().strip().splitlines()[-1]) if os.path.exists(p) else None


def roof_cell(r):
    if r["bound"] == "mfma":
        s = f"MFMA-bound (AI {r['arithmetic_intensity']:.0f}): {r['achieved']:.0f} TF = **{100 * r['frac']:.1f} %** of 2.5 PF, {r['avg_launch_us']:.1f} µs"
        if r.get("shader_clock_mhz"):
            s += f"; {r['shader_clock_mhz']:.0f} MHz held -> {100 * r['frac_of_mfma_peak_at_held_clock']:.1f} % of the peak at that clock"
        return s
    return (f"HBM-bound (AI {r['arithmetic_intensity']:.0f} < {r['ridge']:.0f}): {r['achieved']:.0f} GB/s = {100 * r['frac']:.1f} % of 8 TB/s, "
            f"{r['avg_launch_us']:.1f} µs")


def main():
    cur = sys.argv[1]
    prev = sys.argv[2] if len(sys.argv) > 2 else None
    print("| config | arithmetic | samples/s | no teacher forcing | eval forward (samples/s) | ms/step | launches/step | previous round (samples/s / step) | "
          "forward propagation (dominant kernel), dispatch-attached HIP events in real steps | fabric-port bytes per launch of that kernel vs algorithmic | "
          "whole step through the fabric ports |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    lines = {}
    for name, label in CFGS:
        d = load(cur, name)
        if d is None:
            continue
        lines[name] = d
        p = load(prev, name) if prev else None
        r = d["roofline"]
        tr = r.get("traffic")
        trc = f"{tr / 1e6:.1f} MB vs {r['alg_bytes_per_launch'] / 1e6:.1f} MB ({tr / r['alg_bytes_per_launch']:.2f}×)" if tr else "-"
        st = f"{d['step_fabric_gb']:.1f} GB = {d['step_fabric_tbs']:.2f} TB/s" if d.get("step_fabric_gb") else "-"
        print(f"| {label} | {d['dtype']} | **{d['value']:.0f}** | {d.get('value_no_teacher', 0):.0f} | {d.get('eval_samples_per_s', 0):.0f} | "
              f"{d['ms_per_step']:.2f} | {d.get('kernel_launches_per_step', '-')} | "
              f"{(str(round(p['value'])) + ' / ' + format(p['ms_per_step'], '.2f') + ' ms') if p else '-'} | {roof_cell(r)} | {trc} | {st} |")
    d = lines.get("metrla")
    if d:
        print("\nLegs of the METR-LA (default, driver-run) line:\n")
        for k, what in (("secondary", "EXPY-TKY shape, **bf16 mode**, tolerance 1e-2"),
                        ("secondary_parity", "the same shape, **bf16x3**, tolerance 1e-4; hi/lo operand pairs, one K tile of four images"),
                        ("syn8192", "N = 8192, B = 32, bf16 mode, tile table from `profiles/tiles/`")):
            s = d.get(k)
            if not s or "roofline" not in s:
                print(f"* `{k}`: {s}")
                continue
            r = s["roofline"]
            extra = f" ({s['value_no_teacher']:.0f} without teacher forcing, eval forward {s['eval_samples_per_s']:.0f})" if "value_no_teacher" in s else ""
            clk = (f"; {r['shader_clock_mhz']:.0f} MHz held -> {100 * r['frac_of_mfma_peak_at_held_clock']:.1f} % of the peak at that clock"
                   if r.get("shader_clock_mhz") else "")
            print(f"* `{k}` ({what}): {s['value']:.0f} samples/s{extra}, {s['ms_per_step']:.2f} ms/step, forward propagation {r['achieved']:.0f} TF "
                  f"= {100 * r['frac']:.1f} % of 2.5 PF{clk}")
    print("\nTime per GEMM role (HIP events around every launch of the role during real train steps; the propagation roles and every bf16 product "
          "with events attached to the dispatch):\n")
    for name, label in CFGS:
        d = lines.get(name)
        if not d or "gemm_roles" not in d:
            continue
        parts = [f"{k} {v['ms_per_step']:.2f} ms ({v['launches_per_step']} × {v['avg_us']:.0f} µs, {v['alg_tflops']:.0f} TF)" for k, v in d["gemm_roles"].items()]
        print(f"* {label.split()[0]}: " + ", ".join(parts))
    print("\nCPU leg (`cpu_baseline`, `oracle/megacrn_torch_cpu.py` = the reference's ATen op sequence on PyTorch-CPU, best of 8 / 16 / 32 threads, "
          "and the numpy port; bounded samples scaled linearly):\n")
    for name, label in CFGS:
        d = lines.get(name)
        c = d.get("cpu_baseline") if d else None
        if c:
            print(f"* {label.split()[0]}: torch-CPU restatement {c['value']:.3g} samples/s on {c['cores']} threads (sweep {c['thread_sweep']}), "
                  f"{c['value_1thread']:.3g} on one thread (what the reference pins); numpy port, one thread: {c['value_numpy_port_1thread']:.3g}; {c['cpu_model']}")


if __name__ == "__main__":
    main()
