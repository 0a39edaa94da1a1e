#!/usr/bin/env python3
"""Tuning aid: time every GEMM role of a real train step under each forced tile configuration.
Usage (GPU box): python tools/role_sweep.py [config] [batch]"""
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import megacrn_amd  # noqa: E402
from megacrn_amd._lib import lib, check  # noqa: E402
from megacrn_amd.trainer import FlatTrainer  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "metrla"
cfg = bench.CONFIGS[name]
B = int(sys.argv[2]) if len(sys.argv) > 2 else cfg["B"]
dev = torch.device("cuda", 0)
torch.manual_seed(1234)
model = megacrn_amd.MegaCRN(cfg["N"], 1, 1, cfg["T"], cfg["H"], mem_num=cfg["M"], mem_dim=cfg["D"]).to(dev).train()
tr = FlatTrainer(model, scaler_mean=54.4, scaler_std=19.5)
x, yc, y = bench.synth(cfg, B, 1234, dev)
CFG = ["128x128", "64x128", "128x64", "64x64", "32x128", "256x64", "64x256", "auto"]
res = {}
for c in list(range(7)) + [-1]:
    lib.mcrn_set_gemm_cfg(c)
    for _ in range(2):
        tr.train_step(x, yc, y)
    torch.cuda.synchronize()
    row = {}
    for role in range(1, 7):
        best = 1e9
        for rep in range(2):
            check(lib.mcrn_prof_begin(role), "b")
            tr.train_step(x, yc, y)
            ms, n, af, ef = C.c_double(), C.c_longlong(), C.c_double(), C.c_double()
            torch.cuda.synchronize()
            check(lib.mcrn_prof_end(C.byref(ms), C.byref(n), C.byref(af), C.byref(ef)), "e")
            best = min(best, ms.value)
        row[bench.ROLE_NAMES[role]] = round(best, 3)
    row["sum"] = round(sum(row.values()), 3)
    res[CFG[c]] = row
    print(CFG[c], json.dumps(row), flush=True)
lib.mcrn_set_gemm_cfg(-1)
