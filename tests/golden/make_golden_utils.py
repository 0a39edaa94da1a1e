#!/usr/bin/env python3
"""Golden vectors for the trainer rows either side of the hot path, produced by
running the REFERENCE code on CPU.  Build container only (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_utils.py

* ``model/utils.py`` is imported unmodified: ``DataLoader``, ``StandardScaler``, ``masked_{mae,mape,mse}_loss``.
* ``prepare_x_y`` and ``evaluate`` live in ``model/traintest_MegaCRN.py``, which cannot be imported (module-level
  argparse, ``torchsummary``, dataset files).  Their function definitions are pulled out of that file with ``ast``
  at generation time and executed here against stand-in globals (``args``, ``device``, ``data``, ``scaler``, a
  stub model that returns recorded tensors) - the reference's own statements compute every expected value.
Writes ``tests/golden/utils_f32.npz`` (inputs + expected outputs only; no reference source is stored).
"""
import ast
import logging
import os
import sys
import types

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
REF = "/root/reference/model"
sys.path.insert(0, REF)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

import utils as ref_utils  # noqa: E402  (reference model/utils.py)

OUT = os.path.dirname(os.path.abspath(__file__))
SC_MEAN, SC_STD = 54.4, 19.5


def reference_functions(names):
    """Function definitions `names` of model/traintest_MegaCRN.py, compiled from its source at run time."""
    src = open(os.path.join(REF, "traintest_MegaCRN.py")).read()
    tree = ast.parse(src)
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert len(keep) == len(names), [n.name for n in keep]
    mod = ast.Module(body=keep, type_ignores=[])
    ns = {"torch": torch, "np": np, "nn": nn, "masked_mae_loss": ref_utils.masked_mae_loss,
          "masked_mape_loss": ref_utils.masked_mape_loss, "masked_mse_loss": ref_utils.masked_mse_loss}
    exec(compile(mod, "traintest_MegaCRN.py", "exec"), ns)
    return ns


class StubModel:
    """Returns a fixed (output, h_att, query, pos, neg) per call, in call order."""

    def __init__(self, outs):
        self.outs, self.i = outs, 0

    def eval(self):
        return self

    def __call__(self, x, ycov):
        o = self.outs[self.i]
        self.i += 1
        return tuple(torch.from_numpy(a) for a in o)


def main():
    rec = {}
    rng = np.random.default_rng(77)
    # ---------------- loader + scaler (model/utils.py:6-54) ----------------
    S, T, N = 23, 12, 9
    xs = rng.standard_normal((S, T, N, 2))
    ys = rng.standard_normal((S, T, N, 2))
    xs[..., 0] = 50 + 12 * xs[..., 0]
    ys[..., 0] = 50 + 12 * ys[..., 0]
    ys[rng.random(ys.shape[:3]) < 0.15, 0] = 0.0        # missing readings are exact zeros in the raw data
    xs[..., 1] = rng.random((S, T, N))
    ys[..., 1] = rng.random((S, T, N))
    rec["loader:xs"], rec["loader:ys"] = xs, ys
    np.random.seed(5)
    dl = ref_utils.DataLoader(xs, ys, 4, shuffle=True)
    rec["loader:shuffled_x"] = np.concatenate([b[0] for b in dl.get_iterator()])
    rec["loader:shuffled_y"] = np.concatenate([b[1] for b in dl.get_iterator()])
    rec["loader:meta"] = np.array([dl.size, dl.num_batch, 4, 5], np.int64)      # size, batches, batch_size, seed
    dl2 = ref_utils.DataLoader(xs, ys, 4, shuffle=False)
    rec["loader:plain_x_last"] = list(dl2.get_iterator())[-1][0]
    sc = ref_utils.StandardScaler(mean=xs[..., 0].mean(), std=xs[..., 0].std())
    rec["scaler:mean_std"] = np.array([sc.mean, sc.std])
    rec["scaler:x0"] = sc.transform(xs[..., 0])
    rec["scaler:roundtrip"] = sc.inverse_transform(sc.transform(xs[..., 0]))

    # ---------------- prepare_x_y + evaluate (model/traintest_MegaCRN.py:33-99) ----------------
    ns = reference_functions(["prepare_x_y", "evaluate"])
    args = types.SimpleNamespace(input_dim=1, output_dim=1, lamb=0.01, lamb1=0.01)
    scaler = ref_utils.StandardScaler(mean=SC_MEAN, std=SC_STD)
    xe, ye = xs.copy(), ys.copy()
    xe[..., 0] = scaler.transform(xe[..., 0])
    ye[..., 0] = scaler.transform(ye[..., 0])
    ns.update(args=args, device=torch.device("cpu"), scaler=scaler, logger=logging.getLogger("golden"))
    x0, y0, y1 = ns["prepare_x_y"](xe[:4], ye[:4])
    rec["prep:x"], rec["prep:y"] = xe[:4], ye[:4]
    rec["prep:x0"], rec["prep:y0"], rec["prep:y1"] = x0.numpy(), y0.numpy(), y1.numpy()

    B, D = 4, 10
    loader = ref_utils.DataLoader(xe, ye, B, shuffle=False)
    nb = loader.num_batch
    outs = []
    for i, (xb, yb) in enumerate(loader.get_iterator()):
        yb0 = yb[..., :1].astype(np.float32)
        out = (yb0 + 0.3 * rng.standard_normal(yb0.shape)).astype(np.float32)
        q, p, n_ = (rng.standard_normal((B, N, D)).astype(np.float32) for _ in range(3))
        if i == 1:
            p = (q + 0.05 * rng.standard_normal(q.shape)).astype(np.float32)     # triplet hinge partly inactive
        outs.append((out, np.zeros((B, N, D), np.float32), q, p, n_))
    ns["data"] = {"val_loader": loader}
    # per-batch values exactly as evaluate computes them (for the per-launch check of the device kernel)
    per = []
    for (xb, yb), o in zip(loader.get_iterator(), outs):
        _, yb0, _ = ns["prepare_x_y"](xb, yb)
        y_pred = scaler.inverse_transform(torch.from_numpy(o[0]))
        y_true = scaler.inverse_transform(yb0)
        l1 = ref_utils.masked_mae_loss(y_pred, y_true)
        l2 = nn.TripletMarginLoss(margin=1.0)(torch.from_numpy(o[2]), torch.from_numpy(o[3]), torch.from_numpy(o[4]))
        l3 = nn.MSELoss()(torch.from_numpy(o[2]), torch.from_numpy(o[3]))
        row = [(l1 + args.lamb * l2 + args.lamb1 * l3).item(), l1.item(), l2.item(), l3.item()]
        yt, yp = y_true.permute(1, 0, 2, 3), y_pred.permute(1, 0, 2, 3)
        for sl in (slice(None), slice(2, 3), slice(5, 6), slice(11, 12)):
            row += [ref_utils.masked_mae_loss(yp[sl], yt[sl]).item(), ref_utils.masked_mape_loss(yp[sl], yt[sl]).item(),
                    ref_utils.masked_mse_loss(yp[sl], yt[sl]).item()]
        per.append(row)
    rec["eval:per_batch"] = np.array(per, np.float64)   # [loss, l1, l2, l3, (mae, mape, mse) x {all, 3, 6, 12}]

    # the reference's evaluate() itself, capturing what it logs for mode == 'test'
    lines = []
    handler = logging.Handler()
    handler.emit = lambda r: lines.append(r.getMessage())
    ns["logger"].addHandler(handler)
    ns["logger"].setLevel(logging.INFO)
    ns["data"] = {"test_loader": loader}
    mean_loss, _, _ = ns["evaluate"](StubModel(outs), "test")
    vals = []
    for ln in lines:
        parts = ln.replace(",", "").split()
        vals.append([float(parts[parts.index(k) + 1]) for k in ("mae:", "mape:", "rmse:")])
    rec["eval:mean_loss"] = np.float64(mean_loss)
    rec["eval:logged"] = np.array(vals, np.float64)       # rows: overall, 15, 30, 60 min ; 4 decimals as logged
    for i, o in enumerate(outs):
        for nm, a in zip(("output", "query", "pos", "neg"), (o[0], o[2], o[3], o[4])):
            rec[f"eval:{nm}{i}"] = a
    rec["eval:x"], rec["eval:y"] = xe, ye
    rec["eval:meta"] = np.array([B, T, N, D, nb], np.int64)
    np.savez_compressed(os.path.join(OUT, "utils_f32.npz"), **rec)
    print("utils fixtures written:", len(rec), "arrays; mean_loss", mean_loss, "logged", vals)


if __name__ == "__main__":
    main()
