"""Train step of the reference trainer (model/traintest_MegaCRN.py:115-130) on the HIP hot path.

One step = forward (C ABI) -> 3-term loss (:118-125) -> backward (C ABI, gradients written straight
into one flat fp32 bucket) -> ONE all-reduce(sum) of that bucket over RCCL when world_size > 1 ->
clip_grad_norm_(5) + Adam(lr, eps=1e-3) as a single flat HIP kernel pair.  Parameters stay ordinary
``nn.Parameter``s of the reference shapes (views into the flat buffer), so ``state_dict`` round-trips.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn

from ._lib import lib, check, Dims, Params, Grads
from . import dp


def masked_mae_loss(y_pred, y_true):
    """model/utils.py:126-133 (no in-place NaN patching needed: mask.mean()==0 is the only NaN source)."""
    mask = (y_true != 0).float()
    mask = mask / mask.mean()
    loss = torch.abs(y_pred - y_true) * mask
    loss = torch.where(torch.isnan(loss), torch.zeros_like(loss), loss)
    return loss.mean()


class FlatTrainer:
    def __init__(self, model, *, lr=0.01, eps=1e-3, betas=(0.9, 0.999), max_grad_norm=5.0, lamb=0.01,
                 lamb1=0.01, scaler_mean=0.0, scaler_std=1.0, process_group=None, autotune=True, fused_loss=True):
        if model.num_layers != 1:
            raise ValueError("FlatTrainer drives the fused num_layers==1 path")
        self.model = model
        self.lr, self.eps, self.betas, self.max_grad_norm = lr, eps, betas, float(max_grad_norm)
        self.lamb, self.lamb1 = lamb, lamb1
        self.mean, self.std = float(scaler_mean), float(scaler_std)
        self.group = process_group
        self.autotune = autotune
        self.fused_loss = fused_loss
        self.world = dp.world_size(process_group)
        params = list(model._fused_params())
        dev = params[0].device
        if dev.type != "cuda":
            raise RuntimeError("FlatTrainer needs the model on a HIP device (no CPU fallback)")
        self.bucket = dp.FlatBucket(params, process_group)     # flat parameter / gradient buffers + the collective
        self.n, self.flat_p, self.flat_g = self.bucket.n, self.bucket.flat_p, self.bucket.flat_g
        self._gviews = self.bucket.grad_views
        self.m = torch.zeros(self.n, device=dev)
        self.v = torch.zeros(self.n, device=dev)
        self.scratch = torch.zeros(2048, device=dev)
        self.total_norm = torch.zeros(1, device=dev)
        self.loss_scratch = torch.zeros(4104, device=dev)
        self.losses = torch.zeros(4, device=dev)
        self.params = params
        self.step_count = 0
        self.batches_seen = 0
        self._ws = None
        self._dims_key = None
        self.triplet = nn.TripletMarginLoss(margin=1.0)
        self.mse = nn.MSELoss()

    # -- buffers sized for one batch shape, reused across steps
    def _prepare(self, x):
        m = self.model
        key = (x.shape[0], x.shape[1])
        if key != self._dims_key:
            self.d = Dims(x.shape[0], m.num_nodes, x.shape[1], m.horizon, m.input_dim, m.output_dim, m.ycov_dim,
                          m.rnn_units, m.mem_num, m.mem_dim, m.cheb_k, m.precision)
            nb = lib.mcrn_model_workspace_bytes(C.byref(self.d))
            if nb == 0:
                raise ValueError(lib.mcrn_last_error().decode())
            dev = x.device
            self._ws = torch.empty(nb, dtype=torch.uint8, device=dev)
            if self.autotune:      # once per shape: pick the fastest GEMM tile per signature on-device
                check(lib.mcrn_model_autotune(C.byref(self.d), self._ws.data_ptr(), nb,
                                              torch.cuda.current_stream().cuda_stream), "mcrn_model_autotune")
            B, N, To, od, D = self.d.B, self.d.N, self.d.T_out, self.d.output_dim, self.d.mem_dim
            self.out = torch.empty(B, To, N, od, device=dev)
            self.hatt, self.q, self.pos, self.neg = (torch.empty(B, N, D, device=dev) for _ in range(4))
            self.d_out = torch.empty_like(self.out)
            self.d_q = torch.empty_like(self.q)
            self._dims_key = key

    def loss_fn(self, output, query, pos, neg, labels):
        y_pred = output * self.std + self.mean            # scaler.inverse_transform (:118-119)
        y_true = labels * self.std + self.mean
        loss1 = masked_mae_loss(y_pred, y_true)
        loss2 = self.triplet(query, pos, neg)             # pos/neg are detached (:123)
        loss3 = self.mse(query, pos)
        return loss1 + self.lamb * loss2 + self.lamb1 * loss3

    def _check_inputs(self, x, ycov, labels):
        """The C ABI takes raw pointers: everything the reference would reject with a shape error in
        torch.cat / torch.stack (model/MegaCRN.py:42,185,192) is rejected here before any pointer is passed."""
        m = self.model
        dev = self.flat_p.device
        if x.dim() != 4:
            raise ValueError(f"x must be (B, T_in, N, input_dim); got {tuple(x.shape)}")
        B, T_in = x.shape[0], x.shape[1]
        want = {"x": (B, T_in, m.num_nodes, m.input_dim),
                "y_cov": (B, m.horizon, m.num_nodes, m.ycov_dim),
                "labels": (B, m.horizon, m.num_nodes, m.output_dim)}
        out = []
        for name, t in (("x", x), ("y_cov", ycov), ("labels", labels)):
            if not isinstance(t, torch.Tensor):
                raise ValueError(f"{name} must be a tensor")
            if not t.is_cuda:
                raise RuntimeError(f"{name}: megacrn_amd runs on MI355X only (got a {t.device} tensor); no CPU fallback")
            if t.device != dev:
                raise ValueError(f"{name} lives on {t.device}, the model on {dev}")
            if t.dtype != torch.float32:
                raise TypeError(f"{name}: megacrn_amd computes in fp32; got {t.dtype}")
            if tuple(t.shape) != want[name]:
                raise ValueError(f"{name} must have shape {want[name]}; got {tuple(t.shape)}")
            out.append(t.contiguous())
        return out

    def train_step(self, x, ycov, labels):
        """One optimizer step; returns the (device) loss tensor without synchronising."""
        m = self.model
        x, ycov, labels = self._check_inputs(x, ycov, labels)
        self._prepare(x)
        st = torch.cuda.current_stream().cuda_stream
        teacher = m._teacher_flags(labels, self.batches_seen)
        tarr = (C.c_int * m.horizon)(*[int(f) for f in teacher])
        ps = Params(*[p.data_ptr() for p in self.params])
        gs = Grads(*[g.data_ptr() for g in self._gviews])
        nb = self._ws.numel()
        check(lib.mcrn_model_forward(C.byref(self.d), C.byref(ps), x.data_ptr(), ycov.data_ptr(), labels.data_ptr(),
                                     tarr, self._ws.data_ptr(), nb, self.out.data_ptr(), self.hatt.data_ptr(),
                                     self.q.data_ptr(), self.pos.data_ptr(), self.neg.data_ptr(), st),
              "mcrn_model_forward")
        if self.fused_loss:     # 3 HIP launches instead of ~25 torch kernels + an autograd pass
            d = self.d
            check(lib.mcrn_loss_fwd_bwd(d.B, d.T_out, d.N, d.output_dim, d.mem_dim, self.out.data_ptr(),
                                        labels.data_ptr(), self.q.data_ptr(), self.pos.data_ptr(),
                                        self.neg.data_ptr(), self.mean, self.std, self.lamb, self.lamb1, 1.0,
                                        self.loss_scratch.data_ptr(), self.losses.data_ptr(),
                                        self.d_out.data_ptr(), self.d_q.data_ptr(), st), "mcrn_loss_fwd_bwd")
            d_out, d_q, loss = self.d_out, self.d_q, self.losses[0]
        else:
            out_l = self.out.detach().requires_grad_()
            q_l = self.q.detach().requires_grad_()
            loss = self.loss_fn(out_l, q_l, self.pos, self.neg, labels)
            d_out, d_q = torch.autograd.grad(loss, [out_l, q_l])
        check(lib.mcrn_model_backward(C.byref(self.d), C.byref(ps), tarr, d_out.data_ptr(), None, d_q.data_ptr(),
                                      None, None, self._ws.data_ptr(), nb, C.byref(gs), st), "mcrn_model_backward")
        self.bucket.allreduce()                              # the single collective of the step (no-op at world 1)
        self.step_count += 1
        check(lib.mcrn_flat_clip_adam(self.flat_p.data_ptr(), self.flat_g.data_ptr(), self.m.data_ptr(),
                                      self.v.data_ptr(), self.n, self.lr, self.betas[0], self.betas[1], self.eps,
                                      self.step_count, self.max_grad_norm, self.bucket.grad_scale,
                                      self.scratch.data_ptr(), self.total_norm.data_ptr(), st), "mcrn_flat_clip_adam")
        self.batches_seen += 1
        # a fresh tensor for the caller (self.losses is rewritten by the next step) made by an element-wise kernel:
        # .clone() of a device scalar is a hipMemcpyAsync, ~100 us of queue idle per step on this runtime
        return loss.detach() * 1.0 if self.fused_loss else loss.detach()
