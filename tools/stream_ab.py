#!/usr/bin/env python3
"""A/B: train steps issued on the legacy default (null) stream vs on an explicit non-default stream.
Usage: python tools/stream_ab.py [config]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "metrla"
cfg = bench.CONFIGS[name]
prec = "bf16" if cfg["N"] >= 1024 else "bf16x3"
dev = torch.device("cuda", 0)
tr, (x, yc, y) = bench.make_trainer(name, cfg["B"], prec, dev, 0)


def run(n):
    for _ in range(10):
        tr.train_step(x, yc, y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        tr.train_step(x, yc, y)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


n = 40 if cfg["N"] < 4096 else 5
for rep in range(2):
    a = run(n)
    s = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        b = run(n)
    torch.cuda.synchronize()
    print(f"{name}: default stream {a:.3f} ms/step, explicit stream {b:.3f} ms/step", flush=True)
