#!/usr/bin/env python3
"""Launch one GEMM shape repeatedly (for rocprofv3 --pmc): one_gemm.py M N K tA tB cfg [reps] [prec]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import megacrn_amd
from megacrn_amd._lib import lib, check, set_precision
M, N, K, tA, tB, cfg = map(int, sys.argv[1:7])
reps = int(sys.argv[7]) if len(sys.argv) > 7 else 20
set_precision(sys.argv[8] if len(sys.argv) > 8 else "bf16x3")
A = torch.randn((K, M) if tA else (M, K), device="cuda")
B = torch.randn((N, K) if tB else (K, N), device="cuda")
Cc = torch.empty(M, N, device="cuda")
lib.mcrn_set_gemm_cfg(cfg)
st = torch.cuda.current_stream().cuda_stream
for _ in range(reps):
    check(lib.mcrn_gemm_f32(M, N, K, tA, tB, A.data_ptr(), B.data_ptr(), Cc.data_ptr(), 1.0, 0.0, 1, None, st), "g")
torch.cuda.synchronize()
