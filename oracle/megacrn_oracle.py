"""CPU oracle for the MegaCRN encoder/decoder hot path.  TEST INFRASTRUCTURE ONLY.

This file is a numpy restatement of the reference algorithm in
``/root/reference/model/MegaCRN.py`` (forward) plus a hand-derived backward,
the trainer's 3-term loss (``model/traintest_MegaCRN.py:118-125`` with
``model/utils.py:126-133``), ``clip_grad_norm_`` and Adam.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker / reported CPU baseline.  The product path
(``megacrn_amd``) never imports it and has no CPU fallback.

Parity pinning: the reference repository has no tests or golden vectors
(SURVEY.md section 4), so this oracle is pinned against outputs of the reference itself,
imported in the build container by ``tests/golden/make_golden.py`` which
commits the resulting vectors as ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` re-checks the oracle against those files on
every run (fp32 and fp64).

All tensors use the reference layout: x (B,T,N,C), states (B,N,H), supports
(N,N), weights with row index ``k_global*C + c`` (``model/MegaCRN.py:24-27``).
"""
from __future__ import annotations

import numpy as np

# --------------------------------------------------------------------------
# small helpers
# --------------------------------------------------------------------------


def _softmax_last(x):
    m = x.max(axis=-1, keepdims=True)
    e = np.exp(x - m)
    return e / e.sum(axis=-1, keepdims=True)


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def _prop(S, x):
    """einsum('nm,bmc->bnc', S, x)  -- model/MegaCRN.py:25."""
    B, N, C = x.shape
    xm = np.ascontiguousarray(x.transpose(1, 0, 2)).reshape(N, B * C)
    return (S @ xm).reshape(N, B, C).transpose(1, 0, 2)


def _prop_T(S, d):
    """einsum('mn,bmc->bnc', S, d) = S^T applied over the node axis."""
    B, N, C = d.shape
    dm = np.ascontiguousarray(d.transpose(1, 0, 2)).reshape(N, B * C)
    return (S.T @ dm).reshape(N, B, C).transpose(1, 0, 2)


def _outer_nodes(a, b):
    """sum_b a[b] @ b[b]^T  -> (N,N):  dS contribution."""
    B, N, C = a.shape
    am = np.ascontiguousarray(a.transpose(1, 0, 2)).reshape(N, B * C)
    bm = np.ascontiguousarray(b.transpose(1, 0, 2)).reshape(N, B * C)
    return am @ bm.T


# --------------------------------------------------------------------------
# supports (adaptive adjacency)  -- model/MegaCRN.py:169-173
# --------------------------------------------------------------------------


def supports_fwd(We1, We2, Mem):
    E1 = We1 @ Mem                       # :169
    E2 = We2 @ Mem                       # :170
    L1 = E1 @ E2.T                       # :171 logits of g1 ; g2 logits are L1^T (:172)
    g1 = _softmax_last(np.maximum(L1, 0))
    g2 = _softmax_last(np.maximum(L1.T, 0))
    cache = (We1, We2, Mem, E1, E2, L1, g1, g2)
    return g1, g2, cache


def supports_bwd(dg1, dg2, cache):
    """Backward of row-softmax(relu(L)) for both supports."""
    We1, We2, Mem, E1, E2, L1, g1, g2 = cache
    dZ1 = g1 * (dg1 - (dg1 * g1).sum(-1, keepdims=True))
    dL1 = dZ1 * (L1 > 0)
    dZ2 = g2 * (dg2 - (dg2 * g2).sum(-1, keepdims=True))
    dL2 = dZ2 * (L1.T > 0)
    dLs = dL1 + dL2.T                    # gradient w.r.t. L1 = E1 E2^T
    dE1 = dLs @ E2
    dE2 = dLs.T @ E1
    dWe1 = dE1 @ Mem.T
    dWe2 = dE2 @ Mem.T
    dMem = We1.T @ dE1 + We2.T @ dE2
    return dWe1, dWe2, dMem


# --------------------------------------------------------------------------
# AGCN  -- model/MegaCRN.py:16-28
# --------------------------------------------------------------------------


def agcn_fwd(x, supports, W, b, cheb_k, reference_order=False):
    """x (B,N,C) -> (B,N,O).

    The reference builds Chebyshev *matrices* [I, S, 2 S T_{k-1} - T_{k-2}]
    (:20-22, an N^3 matmul per call) and multiplies each by x (:25).  With
    ``reference_order=True`` this function does exactly that (used to pin the
    oracle on small cases); the default is the mathematically identical
    feature recursion x_k = 2 S x_{k-1} - x_{k-2}, which is what the backward
    below differentiates.
    """
    assert cheb_k >= 2, "cheb_k=1 is broken in the reference (SURVEY.md 3.3)"
    xs = []
    if reference_order:
        N = supports[0].shape[0]
        eye = np.eye(N, dtype=x.dtype)
        for S in supports:
            ks = [eye, S]
            for _ in range(2, cheb_k):
                ks.append((2 * S) @ ks[-1] - ks[-2])
            for A in ks:
                xs.append(_prop(A, x))
    else:
        for S in supports:
            ks = [x, _prop(S, x)]
            for _ in range(2, cheb_k):
                ks.append(2 * _prop(S, ks[-1]) - ks[-2])
            xs.extend(ks)
    xg = np.concatenate(xs, axis=-1)      # :26  (B,N,2*cheb_k*C), column = k_global*C + c
    y = xg @ W + b                        # :27
    cache = (x, supports, W, xs, xg, cheb_k)
    return y, cache


def agcn_bwd(dy, cache):
    """Returns dx, [dS1, dS2], dW, db."""
    x, supports, W, xs, xg, K = cache
    B, N, C = x.shape
    dxg = dy @ W.T
    dW = xg.reshape(-1, xg.shape[-1]).T @ dy.reshape(-1, dy.shape[-1])
    db = dy.reshape(-1, dy.shape[-1]).sum(0)
    dx = np.zeros_like(x)
    dS = []
    for s, S in enumerate(supports):
        d = [dxg[..., (s * K + k) * C:(s * K + k + 1) * C].copy() for k in range(K)]
        xk = xs[s * K:(s + 1) * K]
        dSs = np.zeros_like(S)
        for k in range(K - 1, 1, -1):      # x_k = 2 S x_{k-1} - x_{k-2}
            dSs += 2 * _outer_nodes(d[k], xk[k - 1])
            d[k - 1] += 2 * _prop_T(S, d[k])
            d[k - 2] -= d[k]
        dSs += _outer_nodes(d[1], xk[0])  # x_1 = S x_0
        dx += d[0] + _prop_T(S, d[1])
        dS.append(dSs)
    return dx, dS, dW, db


# --------------------------------------------------------------------------
# AGCRNCell  -- model/MegaCRN.py:38-48
# --------------------------------------------------------------------------


def cell_fwd(x, h, supports, p, cheb_k):
    """p = dict(gate_w, gate_b, update_w, update_b).  z resets the state inside
    the candidate, r blends (:44-47)."""
    H = h.shape[-1]
    xs = np.concatenate([x, h], axis=-1)                       # :42
    g, cg = agcn_fwd(xs, supports, p["gate_w"], p["gate_b"], cheb_k)
    zr = _sigmoid(g)                                            # :43
    z, r = zr[..., :H], zr[..., H:]                             # :44
    cand = np.concatenate([x, z * h], axis=-1)                  # :45
    u, cu = agcn_fwd(cand, supports, p["update_w"], p["update_b"], cheb_k)
    hc = np.tanh(u)                                             # :46
    hn = r * h + (1 - r) * hc                                   # :47
    return hn, (x, h, z, r, hc, cg, cu)


def cell_bwd(dhn, cache):
    """Returns dx, dh, [dS1,dS2], grads dict."""
    x, h, z, r, hc, cg, cu = cache
    din = x.shape[-1]
    dr = dhn * (h - hc)
    dhc = dhn * (1 - r)
    dh = dhn * r
    du = dhc * (1 - hc * hc)
    dcand, dS_u, dWu, dbu = agcn_bwd(du, cu)
    dx = dcand[..., :din].copy()
    dzh = dcand[..., din:]
    dz = dzh * h
    dh = dh + dzh * z
    dg = np.concatenate([dz * z * (1 - z), dr * r * (1 - r)], axis=-1)
    dxs, dS_g, dWg, dbg = agcn_bwd(dg, cg)
    dx += dxs[..., :din]
    dh = dh + dxs[..., din:]
    dS = [dS_u[0] + dS_g[0], dS_u[1] + dS_g[1]]
    return dx, dh, dS, dict(gate_w=dWg, gate_b=dbg, update_w=dWu, update_b=dbu)


# --------------------------------------------------------------------------
# memory query  -- model/MegaCRN.py:159-166
# --------------------------------------------------------------------------


def memory_fwd(h, Mem, Wq):
    q = h @ Wq                                   # :160
    att = _softmax_last(q @ Mem.T)               # :161
    val = att @ Mem                              # :162
    # torch.topk(k=2) (:163): indices of the two largest scores, largest first
    ind = np.argsort(-att, axis=-1, kind="stable")[..., :2]
    pos = Mem[ind[..., 0]]                       # :164
    neg = Mem[ind[..., 1]]                       # :165
    return val, q, pos, neg, (h, Mem, Wq, q, att, ind)


def memory_bwd(dval, dq_ext, dpos, dneg, cache):
    """Returns dh, dMem, dWq.  dpos/dneg may be None (the trainer detaches them,
    model/traintest_MegaCRN.py:123-124)."""
    h, Mem, Wq, q, att, ind = cache
    M, D = Mem.shape
    datt = dval @ Mem.T
    dMem = att.reshape(-1, M).T @ dval.reshape(-1, D)
    dsc = att * (datt - (datt * att).sum(-1, keepdims=True))
    dq = dsc @ Mem
    if dq_ext is not None:
        dq = dq + dq_ext
    dMem += dsc.reshape(-1, M).T @ q.reshape(-1, D)
    if dpos is not None:
        np.add.at(dMem, ind[..., 0].reshape(-1), dpos.reshape(-1, D))
    if dneg is not None:
        np.add.at(dMem, ind[..., 1].reshape(-1), dneg.reshape(-1, D))
    dh = dq @ Wq.T
    dWq = h.reshape(-1, h.shape[-1]).T @ dq.reshape(-1, D)
    return dh, dMem, dWq


# --------------------------------------------------------------------------
# whole model  -- model/MegaCRN.py:168-194
# --------------------------------------------------------------------------

PARAM_KEYS_FIXED = ("memory.Memory", "memory.Wq", "memory.We1", "memory.We2",
                    "proj.0.weight", "proj.0.bias")


def _cell_params(P, prefix, i):
    b = f"{prefix}.dcrnn_cells.{i}."
    return dict(gate_w=P[b + "gate.weights"], gate_b=P[b + "gate.bias"],
                update_w=P[b + "update.weights"], update_b=P[b + "update.bias"])


def sampling_threshold(batches_seen, cl_decay_steps):
    """model/MegaCRN.py:146-147."""
    return cl_decay_steps / (cl_decay_steps + np.exp(batches_seen / cl_decay_steps))


def curriculum_flags(horizon, training, use_cl, batches_seen, cl_decay_steps, rng=np.random):
    """One np.random.uniform draw per decoder step, only in training mode
    (model/MegaCRN.py:188-191).  flag[t] = True -> go = labels[:, t]."""
    flags = []
    for _ in range(horizon):
        f = False
        if training and use_cl:
            c = rng.uniform(0, 1)
            f = bool(c < sampling_threshold(batches_seen, cl_decay_steps))
        flags.append(f)
    return flags


def model_fwd(P, x, ycov, labels=None, teacher=None, *, cheb_k=3, num_layers=1, horizon=None):
    """P: dict keyed like the reference state_dict.  teacher: list[bool] of
    length horizon (the curriculum decisions, see ``curriculum_flags``)."""
    dt = x.dtype
    B, T, N, _ = x.shape
    horizon = ycov.shape[1] if horizon is None else horizon
    teacher = [False] * horizon if teacher is None else list(teacher)
    Mem, Wq = P["memory.Memory"], P["memory.Wq"]
    g1, g2, csup = supports_fwd(P["memory.We1"], P["memory.We2"], Mem)
    sup = [g1, g2]
    H = Wq.shape[0]
    # encoder :65-83
    enc_caches = []
    cur = x
    for i in range(num_layers):
        cp = _cell_params(P, "encoder", i)
        h = np.zeros((B, N, H), dt)                       # :50-51, :174
        states, lc = [], []
        for t in range(T):
            h, c = cell_fwd(cur[:, t], h, sup, cp, cheb_k)
            states.append(h)
            lc.append(c)
        enc_caches.append(lc)
        cur = np.stack(states, axis=1)                    # :78
    h_t = cur[:, -1]                                      # :176
    val, q, pos, neg, cmem = memory_fwd(h_t, Mem, Wq)     # :178
    s0 = np.concatenate([h_t, val], axis=-1)              # :179
    hts = [s0] * num_layers                               # :181
    out_dim = P["proj.0.weight"].shape[0]
    go = np.zeros((B, N, out_dim), dt)                    # :182
    Wp, bp = P["proj.0.weight"], P["proj.0.bias"]
    outs, dec_caches = [], []
    for t in range(horizon):                              # :184
        inp = np.concatenate([go, ycov[:, t]], axis=-1)   # :185
        step_c, new = [], []
        for i in range(num_layers):                       # :109-112
            cp = _cell_params(P, "decoder", i)
            hn, c = cell_fwd(inp, hts[i], sup, cp, cheb_k)
            step_c.append(c)
            new.append(hn)
            inp = hn
        hts = new
        go = inp @ Wp.T + bp                              # :186
        outs.append(go)
        dec_caches.append((step_c, inp))
        if teacher[t]:
            go = labels[:, t]                             # :191
    output = np.stack(outs, axis=1)                       # :192
    cache = dict(P=P, csup=csup, sup=sup, enc=enc_caches, cmem=cmem, dec=dec_caches,
                 teacher=teacher, shapes=(B, T, N, H, out_dim, num_layers, horizon),
                 x_shape=x.shape)
    return (output, val, q, pos, neg), cache


def model_bwd(d_output, cache, d_hatt=None, d_query=None, d_pos=None, d_neg=None):
    """Gradients for every parameter (keys as in the state_dict)."""
    P = cache["P"]
    B, T, N, H, out_dim, L, horizon = cache["shapes"]
    dt = d_output.dtype
    G = {k: np.zeros_like(v) for k, v in P.items()}
    dS = [np.zeros((N, N), dt), np.zeros((N, N), dt)]
    Wp = P["proj.0.weight"]
    Hd = Wp.shape[1]
    dhts = [np.zeros((B, N, Hd), dt) for _ in range(L)]
    dgo_next = np.zeros((B, N, out_dim), dt)     # grad flowing into `go` used by step t+1
    for t in range(horizon - 1, -1, -1):
        step_c, hlast = cache["dec"][t]
        dgo = d_output[:, t].copy()
        if not cache["teacher"][t]:
            dgo += dgo_next                      # go fed to step t+1 is this projection
        G["proj.0.weight"] += dgo.reshape(-1, out_dim).T @ hlast.reshape(-1, Hd)
        G["proj.0.bias"] += dgo.reshape(-1, out_dim).sum(0)
        dinp = dgo @ Wp                          # grad wrt top layer output at step t
        new = [None] * L
        for i in range(L - 1, -1, -1):
            dhn = dhts[i] + dinp
            dx, dh, dSc, g = cell_bwd(dhn, step_c[i])
            b = f"decoder.dcrnn_cells.{i}."
            G[b + "gate.weights"] += g["gate_w"]; G[b + "gate.bias"] += g["gate_b"]
            G[b + "update.weights"] += g["update_w"]; G[b + "update.bias"] += g["update_b"]
            dS[0] += dSc[0]; dS[1] += dSc[1]
            new[i] = dh
            dinp = dx
        dhts = new
        dgo_next = dinp[..., :out_dim]           # input was cat(go, ycov)
    ds0 = sum(dhts)                              # every layer started from the same [h_t | val]
    dh_t = ds0[..., :H].copy()
    dval = ds0[..., H:].copy()
    if d_hatt is not None:
        dval += d_hatt
    dh_m, dMem, dWq = memory_bwd(dval, d_query, d_pos, d_neg, cache["cmem"])
    G["memory.Memory"] += dMem
    G["memory.Wq"] += dWq
    dh_t += dh_m
    # encoder backward (top layer's last state only is used, :176)
    dstack = np.zeros((B, T, N, H), dt)
    dstack[:, -1] = dh_t
    for i in range(L - 1, -1, -1):
        lc = cache["enc"][i]
        din = lc[0][0].shape[-1]
        dprev = np.zeros((B, T, N, din), dt)
        dh = np.zeros((B, N, H), dt)
        b = f"encoder.dcrnn_cells.{i}."
        for t in range(T - 1, -1, -1):
            dhn = dh + dstack[:, t]
            dx, dh, dSc, g = cell_bwd(dhn, lc[t])
            G[b + "gate.weights"] += g["gate_w"]; G[b + "gate.bias"] += g["gate_b"]
            G[b + "update.weights"] += g["update_w"]; G[b + "update.bias"] += g["update_b"]
            dS[0] += dSc[0]; dS[1] += dSc[1]
            dprev[:, t] = dx
        dstack = dprev
    dWe1, dWe2, dMem2 = supports_bwd(dS[0], dS[1], cache["csup"])
    G["memory.We1"] += dWe1
    G["memory.We2"] += dWe2
    G["memory.Memory"] += dMem2
    return G, dS


# --------------------------------------------------------------------------
# trainer-side loss, clip, Adam  (model/traintest_MegaCRN.py:118-130)
# --------------------------------------------------------------------------


def masked_mae_fwd_bwd(y_pred, y_true):
    """model/utils.py:126-133.  Returns loss, dloss/dy_pred."""
    mask = (y_true != 0).astype(np.float32)     # `.float()` in the reference, also in f64 runs
    mask = mask / mask.mean()
    diff = y_pred - y_true
    loss = np.abs(diff) * mask
    bad = loss != loss
    loss = np.where(bad, 0, loss)
    g = np.sign(diff) * mask / diff.size
    g = np.where(bad, 0, g)
    return loss.mean(), g.astype(y_pred.dtype)


def triplet_fwd_bwd(a, p, n, margin=1.0, eps=1e-6):
    """nn.TripletMarginLoss(margin=1.0) (p=2, eps=1e-6, mean); grad wrt anchor only
    (pos/neg are detached by the trainer, :123)."""
    dp = a - p + eps
    dn = a - n + eps
    lp = np.sqrt((dp * dp).sum(-1))
    ln = np.sqrt((dn * dn).sum(-1))
    v = lp - ln + margin
    act = v > 0
    loss = np.where(act, v, 0).mean()
    cnt = v.size
    ga = (dp / lp[..., None] - dn / ln[..., None]) * act[..., None] / cnt
    return loss, ga.astype(a.dtype)


def mse_fwd_bwd(a, b):
    d = a - b
    return (d * d).mean(), (2 * d / d.size).astype(a.dtype)


def loss_fwd_bwd(outs, labels, scaler_mean, scaler_std, lamb=0.01, lamb1=0.01):
    """Three-term loss of traintest_MegaCRN.py:118-125.  Returns
    (loss, loss1, loss2, loss3), d_output, d_query."""
    output, _hatt, query, pos, neg = outs
    y_pred = output * scaler_std + scaler_mean          # :118  inverse_transform
    y_true = labels * scaler_std + scaler_mean          # :119
    l1, g1 = masked_mae_fwd_bwd(y_pred, y_true)
    l2, g2 = triplet_fwd_bwd(query, pos, neg)
    l3, g3 = mse_fwd_bwd(query, pos)
    loss = l1 + lamb * l2 + lamb1 * l3
    d_output = (g1 * scaler_std).astype(output.dtype)
    d_query = (lamb * g2 + lamb1 * g3).astype(query.dtype)
    return (loss, l1, l2, l3), d_output, d_query


# --------------------------------------------------------------------------
# evaluation metrics  (model/utils.py:126-160, model/traintest_MegaCRN.py:63-93)
# --------------------------------------------------------------------------


def _masked(y_pred, y_true, fn):
    """The masked_* family of model/utils.py: mask = (true != 0) / mean(mask), NaN -> 0, mean."""
    mask = (y_true != 0).astype(np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        mask = mask / mask.mean()
        loss = fn(y_pred, y_true) * mask
    loss = np.where(loss != loss, 0, loss)
    return float(loss.mean())


def masked_mae(y_pred, y_true):
    return _masked(y_pred, y_true, lambda p, t: np.abs(p - t))                 # :126-133


def masked_mape(y_pred, y_true):
    return _masked(y_pred, y_true, lambda p, t: np.abs((t - p) / t))           # :135-142


def masked_mse(y_pred, y_true):
    return _masked(y_pred, y_true, lambda p, t: (t - p) ** 2)                  # :153-160


def eval_batch(outs, labels, scaler_mean, scaler_std, lamb=0.01, lamb1=0.01, horizons=(3, 6, 12)):
    """One iteration of evaluate()'s loop (model/traintest_MegaCRN.py:61-88): returns
    [loss, loss1, loss2, loss3, (mae, mape, mse) for the whole batch and for each single-step slice]."""
    output, _hatt, query, pos, neg = outs
    y_pred = (output * np.float32(scaler_std) + np.float32(scaler_mean)).astype(np.float32)
    y_true = (labels * np.float32(scaler_std) + np.float32(scaler_mean)).astype(np.float32)
    l1 = masked_mae(y_pred, y_true)
    l2 = float(triplet_fwd_bwd(query, pos, neg)[0])
    l3 = float(mse_fwd_bwd(query, pos)[0])
    row = [l1 + lamb * l2 + lamb1 * l3, l1, l2, l3]
    for sl in [slice(None)] + [slice(h - 1, h) for h in horizons if h <= y_true.shape[1]]:
        row += [masked_mae(y_pred[:, sl], y_true[:, sl]), masked_mape(y_pred[:, sl], y_true[:, sl]),
                masked_mse(y_pred[:, sl], y_true[:, sl])]
    return row


def eval_epoch(rows):
    """:89-93: means of the per-batch values; RMSE = sqrt of the mean batch MSE."""
    a = np.asarray(rows, np.float64).mean(0)
    res = [a[0]]
    for s in range((a.size - 4) // 3):
        res += [a[4 + 3 * s], a[5 + 3 * s], float(np.sqrt(a[6 + 3 * s]))]
    return res            # [mean loss, (mae, mape, rmse) x slices]


def clip_grad_norm(G, max_norm=5.0):
    """torch.nn.utils.clip_grad_norm_ (:129): returns total norm, scales in place."""
    tot = np.sqrt(sum(float((g.astype(np.float64) ** 2).sum()) for g in G.values()))
    coef = min(1.0, max_norm / (tot + 1e-6))
    for k in G:
        G[k] = (G[k] * G[k].dtype.type(coef))
    return tot


class Adam:
    """torch.optim.Adam(lr, eps) defaults betas=(0.9,0.999), no weight decay (:104)."""

    def __init__(self, P, lr=0.01, eps=1e-3, b1=0.9, b2=0.999):
        self.lr, self.eps, self.b1, self.b2 = lr, eps, b1, b2
        self.m = {k: np.zeros_like(v) for k, v in P.items()}
        self.v = {k: np.zeros_like(v) for k, v in P.items()}
        self.t = 0

    def step(self, P, G):
        self.t += 1
        bc1 = 1 - self.b1 ** self.t
        bc2 = 1 - self.b2 ** self.t
        for k in P:
            g = G[k]
            self.m[k] = self.b1 * self.m[k] + (1 - self.b1) * g
            self.v[k] = self.b2 * self.v[k] + (1 - self.b2) * g * g
            denom = np.sqrt(self.v[k]) / np.sqrt(bc2) + self.eps
            P[k] = (P[k] - (self.lr / bc1) * self.m[k] / denom).astype(P[k].dtype)


def train_step(P, opt, x, ycov, labels, teacher, scaler_mean, scaler_std, *, cheb_k=3,
               num_layers=1, lamb=0.01, lamb1=0.01, max_grad_norm=5.0):
    """One optimizer step of traintest_MegaCRN.py:115-130 (pos/neg detached)."""
    outs, cache = model_fwd(P, x, ycov, labels, teacher, cheb_k=cheb_k, num_layers=num_layers)
    losses, d_out, d_q = loss_fwd_bwd(outs, labels, scaler_mean, scaler_std, lamb, lamb1)
    G, _ = model_bwd(d_out, cache, d_query=d_q)
    gnorm = clip_grad_norm(G, max_grad_norm)
    opt.step(P, G)
    return losses, gnorm, G


# --------------------------------------------------------------------------
# parameter init with the reference scheme (shapes only; RNG differs from torch)
# --------------------------------------------------------------------------


def init_params(num_nodes, input_dim=1, output_dim=1, rnn_units=64, num_layers=1, cheb_k=3,
                ycov_dim=1, mem_num=20, mem_dim=64, seed=0, dtype=np.float32):
    """xavier_normal_ weights, zero biases (model/MegaCRN.py:11-14,149-157,144)."""
    rng = np.random.default_rng(seed)

    def xavier(shape):
        fan_out, fan_in = shape[0], shape[1]
        std = np.sqrt(2.0 / (fan_in + fan_out))
        return (rng.standard_normal(shape) * std).astype(dtype)

    P = {}
    P["memory.Memory"] = xavier((mem_num, mem_dim))
    P["memory.Wq"] = xavier((rnn_units, mem_dim))
    P["memory.We1"] = xavier((num_nodes, mem_num))
    P["memory.We2"] = xavier((num_nodes, mem_num))
    Hd = rnn_units + mem_dim

    def cells(prefix, din0, H):
        for i in range(num_layers):
            din = din0 if i == 0 else H
            C = din + H
            b = f"{prefix}.dcrnn_cells.{i}."
            P[b + "gate.weights"] = xavier((2 * cheb_k * C, 2 * H))
            P[b + "gate.bias"] = np.zeros(2 * H, dtype)
            P[b + "update.weights"] = xavier((2 * cheb_k * C, H))
            P[b + "update.bias"] = np.zeros(H, dtype)

    cells("encoder", input_dim, rnn_units)
    cells("decoder", output_dim + ycov_dim, Hd)
    bound = 1.0 / np.sqrt(Hd)
    P["proj.0.weight"] = rng.uniform(-bound, bound, (output_dim, Hd)).astype(dtype)
    P["proj.0.bias"] = rng.uniform(-bound, bound, (output_dim,)).astype(dtype)
    return P
