#!/usr/bin/env python3
"""Ablation of the bf16x3 GEMM main loop: per-K-tile slope with parts of the loop disabled."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import megacrn_amd
from megacrn_amd._lib import lib, set_precision
from tools.gemm_probe import run
set_precision("bf16x3")
names = {0: "full", 1: "no-mfma", 2: "no-gload", 4: "no-store", 8: "no-barrier", 6: "no-gload+store", 7: "only-barrier", 3: "no-mfma,no-gload", 5: "no-mfma,no-store", 15: "empty-loop"}
for (M, N, tag) in ((13248, 128, "wp-like"), (207, 8448, "prop-like")):
    for cfg in (3, 0):
        for bits, nm in names.items():
            lib.mcrn_set_debug(bits)
            a = run(M, N, 256, 0, 0, cfg, reps=20); b = run(M, N, 2048, 0, 0, cfg, reps=20)
            print(f"{tag:10s} {['128x128','','','64x64'][cfg]:8s} {nm:18s} K=256 {a:6.1f}us K=2048 {b:6.1f}us  slope {(b-a)/56*1e3:6.0f} ns/tile", flush=True)
lib.mcrn_set_debug(0)
