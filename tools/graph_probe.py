#!/usr/bin/env python3
"""Experiment: capture one whole train step (fixed curriculum flags / Adam step) in a HIP graph and replay it."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, megacrn_amd
from megacrn_amd._lib import lib
from megacrn_amd.trainer import FlatTrainer
if len(sys.argv) > 1 and sys.argv[1] == "noside":      # single stream: the captured graph is one linear chain
    lib.mcrn_set_side_stream(0)
cfg = bench.CONFIGS["metrla"]
dev = torch.device("cuda", 0)
torch.manual_seed(1234)
m = megacrn_amd.MegaCRN(cfg["N"], 1, 1, cfg["T"], cfg["H"], mem_num=cfg["M"], mem_dim=cfg["D"]).to(dev).train()
m._teacher_flags = lambda labels, bs: [i % 2 == 0 for i in range(cfg["T"])]
tr = FlatTrainer(m, scaler_mean=54.4, scaler_std=19.5)
x, yc, y = bench.synth(cfg, cfg["B"], 1, dev)
for _ in range(5):
    tr.train_step(x, yc, y)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(20):
    tr.train_step(x, yc, y)
torch.cuda.synchronize()
print("eager  ms/step", (time.perf_counter() - t) / 20 * 1e3)
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    tr.train_step(x, yc, y)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
try:
    with torch.cuda.graph(g):
        tr.train_step(x, yc, y)
    torch.cuda.synchronize()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    print("graph  ms/step", (time.perf_counter() - t) / 20 * 1e3)
except Exception as e:
    print("capture failed:", repr(e)[:300])
