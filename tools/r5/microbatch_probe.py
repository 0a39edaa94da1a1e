"""Probe (round 5): do two half-batch recurrences on two HIP streams beat one full-batch recurrence?
Samples are independent through the whole forward / backward; the small-graph step is a chain of ~240 dependent
launches of 13 - 40 us that fill half the chip, so two such chains side by side might overlap.  Timing only (forward + backward
through the C ABI with a fixed output gradient; no loss / optimizer): full batch on one stream vs two halves on two streams vs two
halves back to back on one stream."""
import ctypes as C
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch
import megacrn_amd
from megacrn_amd._lib import lib, check, Dims, Params, Grads, PRECISIONS
import bench

cfgname = sys.argv[1] if len(sys.argv) > 1 else "metrla"
nsplit = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cfg = bench.CONFIGS[cfgname]
prec = bench.DEFAULT_PREC[cfgname]
dev = torch.device("cuda", 0)
torch.manual_seed(1234)
model = megacrn_amd.MegaCRN(num_nodes=cfg["N"], input_dim=1, output_dim=1, horizon=cfg["T"], rnn_units=cfg["H"],
                            mem_num=cfg["M"], mem_dim=cfg["D"]).to(dev).train()
model.precision = PRECISIONS[prec]
params = list(model._fused_params())
B, T, N, D = cfg["B"], cfg["T"], cfg["N"], cfg["D"]
x, ycov, y = bench.synth(cfg, B, 1234, dev)
teacher = (C.c_int * T)(*([1] * T))


class Part:
    def __init__(self, b0, b1, stream):
        self.b0, self.b1, self.stream = b0, b1, stream
        nb_ = b1 - b0
        self.d = Dims(nb_, N, T, T, 1, 1, 1, cfg["H"], cfg["M"], cfg["D"], 3, model.precision)
        self.nbytes = lib.mcrn_model_workspace_bytes(C.byref(self.d))
        self.ws = torch.empty(self.nbytes, dtype=torch.uint8, device=dev)
        check(lib.mcrn_model_autotune(C.byref(self.d), self.ws.data_ptr(), self.nbytes, torch.cuda.current_stream().cuda_stream), "autotune")
        self.x, self.yc, self.y = x[b0:b1].contiguous(), ycov[b0:b1].contiguous(), y[b0:b1].contiguous()
        self.out = torch.empty(nb_, T, N, 1, device=dev)
        self.o4 = [torch.empty(nb_, N, D, device=dev) for _ in range(4)]
        self.dout = torch.full((nb_, T, N, 1), 1e-3, device=dev)
        self.dq = torch.full((nb_, N, D), 1e-4, device=dev)
        self.grads = [torch.zeros_like(p) for p in params]

    def fwd(self):
        ps = Params(*[p.data_ptr() for p in params])
        check(lib.mcrn_model_forward(C.byref(self.d), C.byref(ps), self.x.data_ptr(), self.yc.data_ptr(), self.y.data_ptr(), teacher,
                                     self.ws.data_ptr(), self.nbytes, self.out.data_ptr(), *[t.data_ptr() for t in self.o4],
                                     self.stream.cuda_stream), "fwd")

    def bwd(self):
        ps = Params(*[p.data_ptr() for p in params])
        gs = Grads(*[g.data_ptr() for g in self.grads])
        check(lib.mcrn_model_backward(C.byref(self.d), C.byref(ps), teacher, self.dout.data_ptr(), None, self.dq.data_ptr(), None, None,
                                      self.ws.data_ptr(), self.nbytes, C.byref(gs), self.stream.cuda_stream), "bwd")


def timeit(fn, n=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


s_main = torch.cuda.current_stream()
streams = [torch.cuda.Stream() for _ in range(nsplit)]
full = Part(0, B, s_main)
bounds = [B * i // nsplit for i in range(nsplit + 1)]
same = [Part(bounds[i], bounds[i + 1], s_main) for i in range(nsplit)]
para = [Part(bounds[i], bounds[i + 1], streams[i]) for i in range(nsplit)]


def run_full():
    full.fwd(); full.bwd()


def run_seq():
    for p in same: p.fwd()
    for p in same: p.bwd()


def run_par():
    for p in para: p.fwd()
    for p in para: p.bwd()


def run_par_fwd_only():
    for p in para: p.fwd()


def run_full_fwd_only():
    full.fwd()


for name, fn in (("full batch, one stream", run_full), (f"{nsplit} parts, one stream", run_seq), (f"{nsplit} parts, {nsplit} streams", run_par),
                 ("full batch forward only", run_full_fwd_only), (f"{nsplit} parts forward only, {nsplit} streams", run_par_fwd_only)):
    ms = timeit(fn)
    print(f"{cfgname} {name:40s} {ms:8.3f} ms  ({B / ms * 1e3:8.1f} samples/s fwd+bwd)", flush=True)
# consistency: gradients of the parts sum to the full batch's
run_full(); run_par(); torch.cuda.synchronize()
for k, (gf, gp) in enumerate(zip(full.grads, zip(*[p.grads for p in para]))):
    s = sum(gp)
    err = float((s - gf).abs().max() / gf.abs().max().clamp_min(1e-30))
    if err > 1e-4:
        print("grad", k, "relerr", err)
print("gradient additivity checked")
