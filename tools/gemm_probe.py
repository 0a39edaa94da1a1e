#!/usr/bin/env python3
"""Tuning aid: time the library GEMM on one shape family to separate fixed from per-K-tile cost."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import megacrn_amd
from megacrn_amd._lib import lib, check, set_precision

def run(M, N, K, tA, tB, cfg, reps=30):
    A = torch.randn((K, M) if tA else (M, K), device="cuda")
    B = torch.randn((N, K) if tB else (K, N), device="cuda")
    Cc = torch.empty(M, N, device="cuda")
    lib.mcrn_set_gemm_cfg(cfg)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        check(lib.mcrn_gemm_f32(M, N, K, tA, tB, A.data_ptr(), B.data_ptr(), Cc.data_ptr(), 1.0, 0.0, 1, None, st), "g")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        lib.mcrn_gemm_f32(M, N, K, tA, tB, A.data_ptr(), B.data_ptr(), Cc.data_ptr(), 1.0, 0.0, 1, None, st)
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps

if __name__ != "__main__":
    shapes = []
CFG = ["128x128", "64x128", "128x64", "64x64", "32x128", "256x64", "64x256", "auto"]
shapes = [] if __name__ != "__main__" else [("prop207", 207, 8448, 207, 0, 0), ("prop207_K414", 207, 8448, 414, 0, 0), ("prop207_K32", 207, 8448, 32, 0, 0),
          ("M256", 256, 8448, 207, 0, 0), ("M128", 128, 8448, 207, 0, 0), ("N4224", 207, 4224, 207, 0, 0),
          ("wp", 13248, 128, 680, 0, 0), ("dgrad", 13248, 680, 256, 0, 1), ("dS", 207, 207, 8448, 0, 1),
          ("big1843", 1843, 2112, 1843, 0, 0), ("sq4096", 4096, 4096, 4096, 0, 0)]
for prec in ("bf16x3",):
    set_precision(prec)
    for name, M, N, K, tA, tB in shapes:
        row = []
        for c in list(range(7)) + [-1]:
            us = run(M, N, K, tA, tB, c)
            row.append(f"{CFG[c]}={us:8.1f}us/{2*M*N*K/us/1e6:7.1f}TF")
        print(prec, f"{name:14s}", "  ".join(row), flush=True)
lib.mcrn_set_gemm_cfg(-1)
