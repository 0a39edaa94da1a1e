"""PyTorch-CPU restatement of the reference hot path and train step.  TEST / BASELINE INFRASTRUCTURE ONLY.

SURVEY.md section 8(d) asks for the CPU path timed beside the GPU to be "PyTorch-CPU ops": this file states the
reference's forward with the SAME ATen op sequence it executes (``torch.eye`` + ``torch.matmul`` Chebyshev matrices per
AGCN call, one broadcast ``einsum`` per support matrix, ``cat``, the weight ``einsum``, ``sigmoid`` / ``split`` / ``tanh``
GRU algebra, ``softmax`` / ``topk`` memory head) as plain functions over a dict of tensors keyed by the reference's
``state_dict`` names, and lets autograd produce the backward pass like ``loss.backward()`` does in the reference
trainer.  The numpy oracle (``megacrn_oracle.py``) remains the parity checker; this file exists so that
``bench.py``'s ``cpu_baseline`` times what the reference itself would cost on the host cores (op-for-op, including its
per-call N^3 Chebyshev products and identity propagations).  It is pinned to the same reference-generated goldens
(``tests/test_oracle_golden.py``).  Only ``tests/`` and ``bench.py``'s ``cpu_baseline`` leg import it.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


def agcn(x, supports, W, b, cheb_k):
    """model/MegaCRN.py:16-28 - the Chebyshev matrix set is rebuilt on every call, identity included."""
    mats = []
    for S in supports:
        ks = [torch.eye(S.shape[0], dtype=S.dtype), S]                     # :20
        for _ in range(2, cheb_k):
            ks.append(torch.matmul(2 * S, ks[-1]) - ks[-2])                # :21-22
        mats.extend(ks)                                                    # :23
    xg = torch.cat([torch.einsum("nm,bmc->bnc", A, x) for A in mats], dim=-1)   # :24-26
    return torch.einsum("bni,io->bno", xg, W) + b                          # :27


def cell(x, h, supports, gw, gb, uw, ub, cheb_k):
    """model/MegaCRN.py:38-48."""
    H = h.shape[-1]
    zr = torch.sigmoid(agcn(torch.cat((x, h), dim=-1), supports, gw, gb, cheb_k))     # :42-43
    z, r = torch.split(zr, H, dim=-1)                                      # :44
    hc = torch.tanh(agcn(torch.cat((x, z * h), dim=-1), supports, uw, ub, cheb_k))    # :45-46
    return r * h + (1 - r) * hc                                            # :47


def forward(P, x, ycov, labels=None, teacher=None, cheb_k=3, num_layers=1):
    """model/MegaCRN.py:168-194.  `teacher[t]` replaces the numpy curriculum draw of :188-190 (same meaning as in the
    numpy oracle and the C ABI).  Returns (output, h_att, query, pos, neg)."""
    Mem, Wq = P["memory.Memory"], P["memory.Wq"]
    E1, E2 = torch.matmul(P["memory.We1"], Mem), torch.matmul(P["memory.We2"], Mem)   # :169-170
    g1 = F.softmax(F.relu(torch.mm(E1, E2.T)), dim=-1)                     # :171
    g2 = F.softmax(F.relu(torch.mm(E2, E1.T)), dim=-1)                     # :172
    sup = [g1, g2]
    B, T_in, N, _ = x.shape
    H = Wq.shape[0]

    def cw(side, i):
        k = f"{side}.dcrnn_cells.{i}."
        return P[k + "gate.weights"], P[k + "gate.bias"], P[k + "update.weights"], P[k + "update.bias"]

    cur = x                                                                # encoder :65-83
    for i in range(num_layers):
        h = torch.zeros(B, N, H, dtype=x.dtype)                            # :50-51,174
        states = []
        for t in range(T_in):
            h = cell(cur[:, t], h, sup, *cw("encoder", i), cheb_k)
            states.append(h)
        cur = torch.stack(states, dim=1)                                   # :78
    h_t = cur[:, -1]                                                       # :176
    query = torch.matmul(h_t, Wq)                                          # :160
    att = torch.softmax(torch.matmul(query, Mem.t()), dim=-1)              # :161
    value = torch.matmul(att, Mem)                                         # :162
    _, ind = torch.topk(att, k=2, dim=-1)                                  # :163
    pos, neg = Mem[ind[:, :, 0]], Mem[ind[:, :, 1]]                        # :164-165
    ht = [torch.cat([h_t, value], dim=-1)] * num_layers                    # :179,181
    T_out = ycov.shape[1]
    od = P["proj.0.weight"].shape[0]
    go = torch.zeros(B, N, od, dtype=x.dtype)                              # :182
    outs = []
    for t in range(T_out):                                                 # :184-191
        inp = torch.cat([go, ycov[:, t]], dim=-1)
        new = []
        for i in range(num_layers):                                        # decoder :103-113
            inp = cell(inp, ht[i], sup, *cw("decoder", i), cheb_k)
            new.append(inp)
        ht = new
        go = F.linear(inp, P["proj.0.weight"], P["proj.0.bias"])           # :186
        outs.append(go)
        if teacher is not None and teacher[t]:
            go = labels[:, t]                                              # :191
    return torch.stack(outs, dim=1), value, query, pos, neg               # :192-194


def loss_terms(outs, labels, mean, std, lamb=0.01, lamb1=0.01):
    """model/traintest_MegaCRN.py:118-125 with model/utils.py:126-133."""
    output, _, query, pos, neg = outs
    y_pred, y_true = output * std + mean, labels * std + mean
    mask = (y_true != 0).float()
    mask = mask / mask.mean()
    l1 = torch.abs(y_pred - y_true) * mask
    l1 = torch.where(torch.isnan(l1), torch.zeros_like(l1), l1).mean()
    l2 = F.triplet_margin_loss(query, pos.detach(), neg.detach(), margin=1.0)
    l3 = F.mse_loss(query, pos.detach())
    return l1 + lamb * l2 + lamb1 * l3


def make_params(P_np, dtype=torch.float32):
    return {k: torch.tensor(np.asarray(v), dtype=dtype, requires_grad=True) for k, v in P_np.items()}


def train_step(P, opt, x, ycov, labels, teacher, mean, std, cheb_k=3, num_layers=1, max_norm=5.0):
    """model/traintest_MegaCRN.py:115-130: zero_grad, forward, 3-term loss, backward, clip_grad_norm_, Adam step."""
    opt.zero_grad()
    loss = loss_terms(forward(P, x, ycov, labels, teacher, cheb_k, num_layers), labels, mean, std)
    loss.backward()
    torch.nn.utils.clip_grad_norm_(list(P.values()), max_norm)
    opt.step()
    return float(loss.item())
