#!/usr/bin/env python3
"""Experiment: do two independent half-batch train chains on two HIP streams overlap on one MI355X?"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, megacrn_amd
from megacrn_amd.trainer import FlatTrainer

cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "metrla"]
dev = torch.device("cuda", 0)

def make(B):
    torch.manual_seed(1234)
    m = megacrn_amd.MegaCRN(cfg["N"], 1, 1, cfg["T"], cfg["H"], mem_num=cfg["M"], mem_dim=cfg["D"]).to(dev).train()
    return FlatTrainer(m, scaler_mean=54.4, scaler_std=19.5), bench.synth(cfg, B, 1, dev)

def run(chains, steps=15):
    streams = [torch.cuda.Stream() for _ in chains]
    for _ in range(4):
        for (tr, d), s in zip(chains, streams):
            with torch.cuda.stream(s):
                tr.train_step(*d)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(steps):
        for (tr, d), s in zip(chains, streams):
            with torch.cuda.stream(s):
                tr.train_step(*d)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / steps

B = cfg["B"]
one = run([make(B)])
print(f"1 chain  x B={B}: {one*1e3:.2f} ms/step  {B/one:.0f} samples/s")
for n in (2, 4):
    t = run([make(B // n) for _ in range(n)])
    print(f"{n} chains x B={B//n}: {t*1e3:.2f} ms/round {B/t:.0f} samples/s")
t = run([make(B) for _ in range(2)])
print(f"2 chains x B={B}: {t*1e3:.2f} ms/round {2*B/t:.0f} samples/s")
