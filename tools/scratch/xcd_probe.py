#!/usr/bin/env python3
"""A/B of the XCD-aware tile order in the bf16x3 GEMM on propagation-shaped problems (debug bit 16 = old order)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import megacrn_amd
from megacrn_amd._lib import lib, set_precision
from tools.gemm_probe import run
set_precision("bf16x3")
for (N, cols, tag) in ((207, 2176 * 2, "METR-LA-ish"), (325, 4352, "PEMS-BAY dec prop"), (1843, 2176, "EXPY-TKY dec prop"), (1843, 1152, "EXPY-TKY enc prop"),
                       (8192, 4224, "N=8192 dec prop"), (4096, 4096, "4096^3")):
    for old in (16, 0):
        lib.mcrn_set_debug(old)
        res = [(run(N, cols, N, 0, 0, c, reps=10), c) for c in range(7)]
        best, c = min(res)
        print(f"{tag:20s} {'old order' if old else 'xcd order'} best cfg {c} {best:9.1f} us  {2.0*N*N*cols/best/1e6:7.1f} TF   all: " + " ".join(f"{t:.0f}" for t, _ in res), flush=True)
lib.mcrn_set_debug(0)
