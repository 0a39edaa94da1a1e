"""debug: reproduce the intermittent decoder-gate weight-gradient mismatch (MCRN_DS_MERGE=0) and print where it is."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
import megacrn_amd as amd
from oracle import megacrn_oracle as O
from helpers import load_case, relerr
import test_gpu_parity as T

amd._lib.set_precision("bf16x3"); amd.test_precision = amd._lib.PRECISIONS["bf16x3"]
dev = T.dev
for name in T.CASES:      # history: the golden train steps first, like the failing selection
    rec, P, m = load_case(name, "f32")
    model = T.build(amd, P, m).train()
    teacher = [bool(v) for v in rec["teacher"]]
    model._teacher_flags = lambda labels, bs: teacher
    outs = model(dev(rec["x"]), dev(rec["ycov"]), dev(rec["labels"]), int(rec["batches_seen"]))
    sum(o.sum() for o in outs[:3]).backward()
    torch.cuda.synchronize()
N, B, H, D, Tn = 250, 320, 16, 8, 1
M, cheb_k = 4, 3
P = O.init_params(N, rnn_units=H, mem_num=M, mem_dim=D, cheb_k=cheb_k, seed=5)
rng = np.random.default_rng(9)
for k in P:
    if k.endswith("bias"):
        P[k] = (0.05 * rng.standard_normal(P[k].shape)).astype(np.float32)
x = rng.standard_normal((B, Tn, N, 1)).astype(np.float32)
ycov = rng.random((B, Tn, N, 1)).astype(np.float32)
y = rng.standard_normal((B, Tn, N, 1)).astype(np.float32)
teacher = [False, True][:Tn]
m = dict(N=N, T_out=Tn, H=H, num_layers=1, cheb_k=cheb_k, M=M, D=D, cl_decay=2000)
wts = None
ref = None
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    model = T.build(amd, P, m).train()
    model._teacher_flags = lambda labels, bs: teacher
    outs = model(dev(x), dev(ycov), dev(y), 0)
    if wts is None:
        wts = [rng.standard_normal(o.shape) for o in outs[:3]]
    sum((o * dev(w)).sum() for o, w in zip(outs[:3], wts)).backward()
    torch.cuda.synchronize()
    g = dict(model.named_parameters())["decoder.dcrnn_cells.0.gate.weights"].grad.cpu().numpy()
    if ref is None:
        P64 = {k: v.astype(np.float64) for k, v in P.items()}
        o64, cache = O.model_fwd(P64, x.astype(np.float64), ycov.astype(np.float64), y.astype(np.float64), teacher, cheb_k=cheb_k)
        G, _ = O.model_bwd(wts[0], cache, d_hatt=wts[1], d_query=wts[2])
        ref = G["decoder.dcrnn_cells.0.gate.weights"]
    err = np.abs(g - ref) / np.abs(ref).max()
    bad = np.argwhere(err > 1e-4)
    print(f"rep {rep}: relerr {err.max():.3e}  shape {g.shape}  bad elements {len(bad)}", flush=True)
    if len(bad):
        rows = sorted(set(int(b[0]) for b in bad)); cols = sorted(set(int(b[1]) for b in bad))
        print("   bad rows", rows[:40], "... cols", cols[:50])
        print("   ratio got/ref at first bad:", [(tuple(b), float(g[tuple(b)]), float(ref[tuple(b)])) for b in bad[:5]])
