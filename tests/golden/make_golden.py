#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE implementation on CPU.

Run in the build container only (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

It imports ``/root/reference/model/MegaCRN.py`` and ``model/utils.py``
unmodified, runs seeded cases and writes ``tests/golden/*.npz`` (inputs,
state_dict, outputs, losses, parameter gradients, one Adam step, a 3-step loss
trajectory).  The .npz files are data; no reference source is stored.  The GPU
box never sees /root/reference - tests there read only the .npz files.
"""
import os
import sys

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
REF = "/root/reference/model"
sys.path.insert(0, REF)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

from MegaCRN import MegaCRN, AGCN, AGCRNCell  # noqa: E402  (reference)
from utils import masked_mae_loss  # noqa: E402  (reference)

OUT = os.path.dirname(os.path.abspath(__file__))
SC_MEAN, SC_STD = 54.4, 19.5   # METR-LA-like scaler


def synth_batch(B, T_in, T_out, N, seed, dtype):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((B, T_in, N, 1))
    y = rng.standard_normal((B, T_out, N, 1))
    miss = (0.0 - SC_MEAN) / SC_STD
    x[rng.random(x.shape) < 0.08] = miss
    y[rng.random(y.shape) < 0.08] = miss
    t0 = rng.integers(0, 288, size=(B, 1, 1, 1))
    ycov = ((t0 + T_in + np.arange(T_out).reshape(1, T_out, 1, 1)) / 288.0) % 1.0
    ycov = np.broadcast_to(ycov, (B, T_out, N, 1)).copy()
    return x.astype(dtype), ycov.astype(dtype), y.astype(dtype)


def loss_terms(model_out, y, lamb=0.01, lamb1=0.01):
    output, h_att, query, pos, neg = model_out
    y_pred = output * SC_STD + SC_MEAN
    y_true = y * SC_STD + SC_MEAN
    loss1 = masked_mae_loss(y_pred, y_true)
    loss2 = nn.TripletMarginLoss(margin=1.0)(query, pos.detach(), neg.detach())
    loss3 = nn.MSELoss()(query, pos.detach())
    return loss1 + lamb * loss2 + lamb1 * loss3, loss1, loss2, loss3


def run_case(name, *, B, N, T_in, T_out, H, M, D, cheb_k=3, num_layers=1, seed=1,
             batches_seen=15200, cl_decay=2000, dtypes=("f32", "f64"), traj=True, slim=False):
    for dn in dtypes:
        tdt = torch.float32 if dn == "f32" else torch.float64
        ndt = np.float32 if dn == "f32" else np.float64
        torch.set_default_dtype(tdt)
        torch.manual_seed(seed)
        model = MegaCRN(num_nodes=N, input_dim=1, output_dim=1, horizon=T_out, rnn_units=H,
                        num_layers=num_layers, cheb_k=cheb_k, mem_num=M, mem_dim=D,
                        cl_decay_steps=cl_decay, use_curriculum_learning=True)
        if dn == "f64":
            model = model.double()
        # xavier biases are zero in the reference; perturb them so bias paths are exercised
        with torch.no_grad():
            for n_, p_ in model.named_parameters():
                if n_.endswith("bias"):
                    p_.add_(0.05 * torch.randn_like(p_))
        sd0 = {k: v.detach().clone().numpy().astype(ndt) for k, v in model.state_dict().items()}
        x, ycov, y = synth_batch(B, T_in, T_out, N, seed + 100, ndt)
        xt, yct, yt = map(torch.from_numpy, (x, ycov, y))
        rec = {}
        for k, v in sd0.items():
            rec["p:" + k] = v
        rec.update(x=x, ycov=ycov, labels=y, batches_seen=np.int64(batches_seen),
                   meta=np.array([B, N, T_in, T_out, H, M, D, cheb_k, num_layers, cl_decay], np.int64))
        cap = {}
        hk = model.encoder.register_forward_hook(
            lambda m, inp, out: cap.update(g1=inp[2][0].detach().numpy().copy(),
                                           g2=inp[2][1].detach().numpy().copy(),
                                           h_en=out[0].detach().numpy().copy()))
        # ---- eval forward (no curriculum) ----
        model.eval()
        with torch.no_grad():
            o = model(xt, yct)
        for nm, t in zip(("output", "h_att", "query", "pos", "neg"), o):
            rec["eval:" + nm] = t.numpy().copy()
        rec["g1"], rec["g2"] = cap["g1"], cap["g2"]
        if not slim:
            rec["eval:h_en"] = cap["h_en"]
        hk.remove()
        # ---- train forward/backward with recorded curriculum draws ----
        model.train()
        np.random.seed(seed + 7)
        thr = model.compute_sampling_threshold(batches_seen)
        draws = np.random.uniform(0, 1, size=T_out)      # same stream the model will consume
        rec["teacher"] = (draws < thr).astype(np.int64)
        np.random.seed(seed + 7)
        model.zero_grad()
        o = model(xt, yct, yt, batches_seen)
        loss, l1, l2, l3 = loss_terms(o, yt)
        loss.backward()
        for nm, t in zip(("output", "h_att", "query", "pos", "neg"), o):
            rec["train:" + nm] = t.detach().numpy().copy()
        rec["train:loss"] = np.array([loss.item(), l1.item(), l2.item(), l3.item()], np.float64)
        for k, p in model.named_parameters():
            rec["g:" + k] = p.grad.detach().numpy().copy()
        opt = torch.optim.Adam(model.parameters(), lr=0.01, eps=1e-3)
        gn = torch.nn.utils.clip_grad_norm_(model.parameters(), 5)
        rec["train:gnorm"] = np.array(float(gn), np.float64)
        opt.step()
        if not slim:
            for k, v in model.state_dict().items():
                rec["p1:" + k] = v.detach().numpy().copy()
        # ---- 3-step loss trajectory (steps 2,3 continue from the step above) ----
        if traj:
            tl = [loss.item()]
            tflags = [rec["teacher"]]
            for s in range(1, 3):
                np.random.seed(seed + 7 + s)
                thr = model.compute_sampling_threshold(batches_seen + s)
                tflags.append((np.random.uniform(0, 1, size=T_out) < thr).astype(np.int64))
                np.random.seed(seed + 7 + s)
                opt.zero_grad()
                o = model(xt, yct, yt, batches_seen + s)
                ls = loss_terms(o, yt)[0]
                ls.backward()
                torch.nn.utils.clip_grad_norm_(model.parameters(), 5)
                opt.step()
                tl.append(ls.item())
            rec["traj:loss"] = np.array(tl, np.float64)
            rec["traj:teacher"] = np.stack(tflags)
        np.savez_compressed(os.path.join(OUT, f"{name}_{dn}.npz"), **rec)
        print(name, dn, "loss", rec["train:loss"], "teacher", rec["teacher"].tolist())
    torch.set_default_dtype(torch.float32)


def run_ops():
    """Stand-alone AGCN / AGCRNCell with generic (non-softmax) supports, f32 + f64."""
    for dn in ("f32", "f64"):
        tdt = torch.float32 if dn == "f32" else torch.float64
        torch.set_default_dtype(tdt)
        rec = {}
        for tag, (B, N, C, O, K) in {"a": (3, 13, 9, 16, 3), "b": (2, 37, 5, 7, 2)}.items():
            torch.manual_seed(11)
            m = AGCN(C, O, K)
            if dn == "f64":
                m = m.double()
            with torch.no_grad():
                m.bias.add_(0.1 * torch.randn_like(m.bias))
            x = torch.randn(B, N, C, requires_grad=True)
            s1 = (0.3 * torch.randn(N, N)).requires_grad_()
            s2 = (0.3 * torch.randn(N, N)).requires_grad_()
            y = m(x, [s1, s2])
            dy = torch.randn_like(y)
            y.backward(dy)
            rec.update({f"agcn_{tag}:meta": np.array([B, N, C, O, K]),
                        f"agcn_{tag}:x": x.detach().numpy(), f"agcn_{tag}:s1": s1.detach().numpy(),
                        f"agcn_{tag}:s2": s2.detach().numpy(), f"agcn_{tag}:w": m.weights.detach().numpy(),
                        f"agcn_{tag}:b": m.bias.detach().numpy(), f"agcn_{tag}:y": y.detach().numpy(),
                        f"agcn_{tag}:dy": dy.numpy(), f"agcn_{tag}:dx": x.grad.numpy(),
                        f"agcn_{tag}:ds1": s1.grad.numpy(), f"agcn_{tag}:ds2": s2.grad.numpy(),
                        f"agcn_{tag}:dw": m.weights.grad.numpy(), f"agcn_{tag}:db": m.bias.grad.numpy()})
        for tag, (B, N, din, H, K) in {"a": (3, 13, 2, 8, 3), "b": (2, 21, 8, 8, 2)}.items():
            torch.manual_seed(12)
            c = AGCRNCell(N, din, H, K)
            if dn == "f64":
                c = c.double()
            with torch.no_grad():
                c.gate.bias.add_(0.1 * torch.randn_like(c.gate.bias))
                c.update.bias.add_(0.1 * torch.randn_like(c.update.bias))
            x = torch.randn(B, N, din, requires_grad=True)
            h = torch.randn(B, N, H, requires_grad=True)
            s1 = torch.softmax(torch.randn(N, N), -1).requires_grad_()
            s2 = torch.softmax(torch.randn(N, N), -1).requires_grad_()
            hn = c(x, h, [s1, s2])
            dh = torch.randn_like(hn)
            hn.backward(dh)
            rec.update({f"cell_{tag}:meta": np.array([B, N, din, H, K]),
                        f"cell_{tag}:x": x.detach().numpy(), f"cell_{tag}:h": h.detach().numpy(),
                        f"cell_{tag}:s1": s1.detach().numpy(), f"cell_{tag}:s2": s2.detach().numpy(),
                        f"cell_{tag}:gw": c.gate.weights.detach().numpy(), f"cell_{tag}:gb": c.gate.bias.detach().numpy(),
                        f"cell_{tag}:uw": c.update.weights.detach().numpy(), f"cell_{tag}:ub": c.update.bias.detach().numpy(),
                        f"cell_{tag}:hn": hn.detach().numpy(), f"cell_{tag}:dhn": dh.numpy(),
                        f"cell_{tag}:dx": x.grad.numpy(), f"cell_{tag}:dh": h.grad.numpy(),
                        f"cell_{tag}:ds1": s1.grad.numpy(), f"cell_{tag}:ds2": s2.grad.numpy(),
                        f"cell_{tag}:dgw": c.gate.weights.grad.numpy(), f"cell_{tag}:dgb": c.gate.bias.grad.numpy(),
                        f"cell_{tag}:duw": c.update.weights.grad.numpy(), f"cell_{tag}:dub": c.update.bias.grad.numpy()})
        np.savez_compressed(os.path.join(OUT, f"ops_{dn}.npz"), **rec)
        print("ops", dn, "written")
    torch.set_default_dtype(torch.float32)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "cheb4":      # (adds the one case without rewriting the others)
        run_case("cheb4", B=2, N=19, T_in=3, T_out=3, H=8, M=5, D=8, cheb_k=4, seed=6)
        sys.exit(0)
    run_ops()
    run_case("tiny", B=3, N=13, T_in=4, T_out=4, H=8, M=5, D=8, seed=1)
    run_case("odd", B=2, N=37, T_in=3, T_out=5, H=12, M=7, D=10, seed=2)
    run_case("layers2", B=2, N=11, T_in=3, T_out=3, H=8, M=5, D=8, num_layers=2, seed=3)
    run_case("cheb2", B=2, N=17, T_in=3, T_out=3, H=8, M=5, D=8, cheb_k=2, seed=4)
    run_case("metrla", B=2, N=207, T_in=12, T_out=12, H=64, M=20, D=64, seed=5, dtypes=("f32",), traj=False, slim=True)
    run_case("cheb4", B=2, N=19, T_in=3, T_out=3, H=8, M=5, D=8, cheb_k=4, seed=6)     # --max_diffusion_step 4 (:21-22 recurse twice)
