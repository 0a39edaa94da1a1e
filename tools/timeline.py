#!/usr/bin/env python3
"""Measurement only: per-phase wall-clock timeline inside the fused propagation kernels.
Needs the instrumented build:  make -C megacrn_amd/csrc timeline [FENCE=2]   then
    MEGACRN_LIB=megacrn_amd/libmegacrn_hip_tl.so python tools/timeline.py [config] [eval|train]
Stamps are the 100 MHz wall clock of thread 0 of each workgroup; the table shows, over the workgroups of the LAST
launch of each kernel kind, the median / max time of every phase in microseconds."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import megacrn_amd  # noqa: E402
from megacrn_amd._lib import lib  # noqa: E402
from megacrn_amd.trainer import FlatTrainer  # noqa: E402

PHASES = {
    # (a workgroup that walks 2 units overwrites its stamps: its "unit setup" then contains the whole first unit)
    0: ["S frags", "unit setup", "stage(load+cvt+lds)", "x0 prefetch + barrier", "mma1 + X1 stores", "barrier", "to_img + barrier", "mma2 + X2 stores", "drain"],
    1: ["S frags", "unit setup", "stage(load+cvt+lds)", "dP[1+2s] prefetch + barrier", "mma1 + dP[1+2s] stores + dP[0] prefetch", "barrier", "to_img + barrier", "mma2 + dP[0] stores", "drain"],
}
name = sys.argv[1] if len(sys.argv) > 1 else "metrla"
mode = sys.argv[2] if len(sys.argv) > 2 else "train"
cfg = bench.CONFIGS[name]
dev = torch.device("cuda", 0)
torch.manual_seed(1234)
model = megacrn_amd.MegaCRN(cfg["N"], 1, 1, cfg["T"], cfg["H"], mem_num=cfg["M"], mem_dim=cfg["D"]).to(dev)
x, yc, y = bench.synth(cfg, cfg["B"], 1234, dev)
if mode == "train":
    tr = FlatTrainer(model.train(), scaler_mean=54.4, scaler_std=19.5)
    for _ in range(5):
        tr.train_step(x, yc, y)
else:
    model.eval()
    with torch.no_grad():
        for _ in range(5):
            model(x, yc)
torch.cuda.synchronize()
buf = np.zeros((10, 512, 12), dtype=np.uint64)
lib.mcrn_debug_timeline.restype = C.c_int
lib.mcrn_debug_timeline.argtypes = [C.c_void_p, C.c_size_t]
assert lib.mcrn_debug_timeline(buf.ctypes.data, buf.nbytes) == 0
for kind, names in PHASES.items():
    t = buf[kind].astype(np.int64)
    live = t[:, 0] > 0
    if not live.any():
        continue
    t = t[live]
    t0 = t[:, 0].min()
    print(f"kind {kind}: {live.sum()} workgroups, first start -> last end {(t[:, 9].max() - t0) / 100:.2f} us, "
          f"start skew (max-min) {(t[:, 0].max() - t0) / 100:.2f} us")
    for i, nm in enumerate(names):
        d = (t[:, i + 1] - t[:, i]) / 100.0
        print(f"   {nm:40s} median {np.median(d):7.2f}  max {d.max():7.2f} us")
    tot = (t[:, 9] - t[:, 0]) / 100.0
    print(f"   {'workgroup total':40s} median {np.median(tot):7.2f}  max {tot.max():7.2f} us")
t = buf[2].astype(np.int64)
t = t[t[:, 0] > 0]
if len(t):
    print(f"kind 2 (ds_small): {len(t)} workgroups, panels/workgroup median {int(np.median(t[:, 9]))}; first start -> last end {(t[:, 3].max() - t[:, 0].min()) / 100:.2f} us")
    for nm, d in (("K loop", t[:, 1] - t[:, 0]), ("epilogue issue", t[:, 2] - t[:, 1]), ("epilogue drain", t[:, 3] - t[:, 2]),
                  ("  sum loop top", t[:, 4]), ("  sum publish next", t[:, 5]), ("  sum fetch issue", t[:, 6]), ("  sum mfma", t[:, 7]), ("  sum barrier", t[:, 8])):
        print(f"   {nm:40s} median {np.median(d) / 100:7.2f}  max {d.max() / 100:7.2f} us")

ROLES = ["misc", "propagate", "weight_pool", "dgrad", "propagate_T", "adjacency_grad", "weight_grad"]
for role in range(7):
    t = buf[3 + role].astype(np.int64)
    nblk = int(t[0, 10]) if t[0, 10] > 0 else 0
    if nblk == 0:
        continue
    t = t[:min(nblk, 512)]
    t = t[t[:, 8] > 0]
    tile = int(t[0, 7])
    print(f"GEMM role {ROLES[role]}: last launch {nblk} workgroups, tile {tile // 1000}x{tile % 1000}, K-tiles {int(np.median(t[:, 6]))}, "
          f"first start -> last end {(t[:, 9].max() - t[:, 8].min()) / 100:.2f} us")
    for i, nm in enumerate(["prologue", "sum load issue", "sum cvt+LDS store", "sum MFMA block", "sum barrier", "epilogue"]):
        print(f"   {nm:40s} median {np.median(t[:, i]) / 100:7.2f}  max {t[:, i].max() / 100:7.2f} us")
    tot = (t[:, 9] - t[:, 8]) / 100.0
    print(f"   {'workgroup total':40s} median {np.median(tot):7.2f}  max {tot.max():7.2f} us")
