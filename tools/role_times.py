#!/usr/bin/env python3
"""Per-role GPU time of one tuned train step (HIP-event role profiler).  Usage: python tools/role_times.py [config]"""
import ctypes as C
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import megacrn_amd  # noqa: E402
from megacrn_amd._lib import lib, check  # noqa: E402
from megacrn_amd.trainer import FlatTrainer  # noqa: E402

for name in (sys.argv[1:] or ["metrla"]):
    cfg = bench.CONFIGS[name]
    B = cfg["B"]
    dev = torch.device("cuda", 0)
    torch.manual_seed(1234)
    model = megacrn_amd.MegaCRN(cfg["N"], 1, 1, cfg["T"], cfg["H"], mem_num=cfg["M"], mem_dim=cfg["D"]).to(dev).train()
    tr = FlatTrainer(model, scaler_mean=54.4, scaler_std=19.5)
    x, yc, y = bench.synth(cfg, B, 1234, dev)
    for _ in range(3):
        tr.train_step(x, yc, y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        tr.train_step(x, yc, y)
    torch.cuda.synchronize()
    step = (time.perf_counter() - t0) * 100
    row = {}
    for role in range(1, 7):
        best, cnt, alg = 1e9, 0, 0.0
        for rep in range(2):
            check(lib.mcrn_prof_begin(role), "b")
            tr.train_step(x, yc, y)
            ms, n, af, ef = C.c_double(), C.c_longlong(), C.c_double(), C.c_double()
            torch.cuda.synchronize()
            check(lib.mcrn_prof_end(C.byref(ms), C.byref(n), C.byref(af), C.byref(ef)), "e")
            if ms.value < best:
                best, cnt, alg = ms.value, n.value, af.value
        row[bench.ROLE_NAMES[role]] = {"ms": round(best, 3), "launches": cnt, "alg_TF": round(alg / best / 1e9, 1) if best > 0 else 0}
    print(name, f"step {step:.2f} ms", json.dumps(row), flush=True)
