#!/usr/bin/env python3
"""Is a train step limited by the host's enqueue rate?  Prints host enqueue time per step (train_step returning,
no sync) next to the synchronised step time, and the launch count.  Usage: python tools/host_bound.py [config]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import megacrn_amd  # noqa: E402
from megacrn_amd._lib import lib  # noqa: E402
from megacrn_amd.trainer import FlatTrainer  # noqa: E402

for name in (sys.argv[1:] or ["metrla"]):
    cfg = bench.CONFIGS[name]
    dev = torch.device("cuda", 0)
    torch.manual_seed(1234)
    prec = "bf16" if cfg["N"] >= 1024 else "bf16x3"           # the arithmetic bench.py runs the configuration in
    tr, (x, yc, y) = bench.make_trainer(name, cfg["B"], prec, dev, 0)
    for _ in range(20):
        tr.train_step(x, yc, y)
    torch.cuda.synchronize()
    n = 50
    t0 = time.perf_counter()
    for _ in range(n):
        tr.train_step(x, yc, y)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    # one step at a time: enqueue, then wait -> GPU time of an isolated step when the queue is never empty at start
    iso = []
    for _ in range(10):
        torch.cuda.synchronize()
        a = time.perf_counter()
        tr.train_step(x, yc, y)
        b = time.perf_counter()
        torch.cuda.synchronize()
        c = time.perf_counter()
        iso.append((b - a, c - a))
    print(f"{name}: host enqueue {1e3*(t1-t0)/n:.2f} ms/step, synchronised {1e3*(t2-t0)/n:.2f} ms/step, "
          f"launches/step {lib.mcrn_last_launch_count()}, isolated step: enqueue {1e3*min(i[0] for i in iso):.2f} ms, total {1e3*min(i[1] for i in iso):.2f} ms",
          flush=True)
