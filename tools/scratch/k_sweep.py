#!/usr/bin/env python3
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import megacrn_amd
from megacrn_amd._lib import lib, check, set_precision
from tools.gemm_probe import run  # noqa
set_precision("bf16x3")
for (M, N, tag) in ((13248, 128, "wp-like"), (207, 8448, "prop-like"), (13248, 680, "dgrad-like(N=680)")):
    for cfg in (3, 0):
        row = []
        for K in (32, 64, 128, 256, 512, 1024, 2048):
            us = run(M, N, K, 0, 0, cfg, reps=20)
            row.append(f"K={K}:{us:6.1f}")
        print(tag, ["128x128","","","64x64"][cfg], " ".join(row), flush=True)
