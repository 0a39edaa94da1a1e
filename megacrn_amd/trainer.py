"""Train step of the reference trainer (model/traintest_MegaCRN.py:115-130) on the HIP hot path.

One step = forward (C ABI) -> 3-term loss (:118-125, one fused HIP kernel group) -> backward (C ABI, gradients written
straight into one flat fp32 bucket) -> ONE all-reduce(sum) of that bucket over RCCL when world_size > 1 ->
clip_grad_norm_(5) + Adam(lr, eps=1e-3) as a single flat HIP kernel pair.  Parameters stay ordinary
``nn.Parameter``s of the reference shapes (views into the flat buffer), so ``state_dict`` round-trips.

``num_layers == 1`` (the reference default, ``--num_rnn_layers 1``) runs the fused model entry points.
``num_layers > 1`` (``model/MegaCRN.py:71-78,109-112``) runs the module's composed path - the reference's loop structure
over the per-cell HIP ops - under autograd, with ``.grad`` of every parameter bound to its slice of the flat gradient
bucket, so the loss kernels, the single all-reduce and the fused clip+Adam tail are the same in both cases.
"""
from __future__ import annotations

import ctypes as C

import torch

from ._lib import lib, check, Dims, Params, Grads
from . import dp


class FlatTrainer:
    def __init__(self, model, *, lr=0.01, eps=1e-3, betas=(0.9, 0.999), max_grad_norm=5.0, lamb=0.01,
                 lamb1=0.01, scaler_mean=0.0, scaler_std=1.0, process_group=None, autotune=True):
        self.model = model
        self.fused = model.num_layers == 1
        self.lr, self.eps, self.betas, self.max_grad_norm = lr, eps, betas, float(max_grad_norm)
        self.lamb, self.lamb1 = lamb, lamb1
        self.mean, self.std = float(scaler_mean), float(scaler_std)
        self.group = process_group
        self.autotune = autotune
        self.world = dp.world_size(process_group)
        params = list(model._fused_params()) if self.fused else list(model.parameters())
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

    # -- buffers sized for one batch shape, reused across steps
    def _prepare(self, x):
        m = self.model
        key = (x.shape[0], x.shape[1])
        if key != self._dims_key:
            self.d = Dims(x.shape[0], m.num_nodes, x.shape[1], m.horizon, m.input_dim, m.output_dim, m.ycov_dim,
                          m.rnn_units, m.mem_num, m.mem_dim, m.cheb_k, m.precision)
            dev = x.device
            B, N, To, od, D = self.d.B, self.d.N, self.d.T_out, self.d.output_dim, self.d.mem_dim
            if self.fused:
                nb = lib.mcrn_model_workspace_bytes(C.byref(self.d))
                if nb == 0:
                    raise ValueError(lib.mcrn_last_error().decode())
                self._ws = torch.empty(nb, dtype=torch.uint8, device=dev)
                if self.autotune:      # once per shape: pick the fastest GEMM tile per signature on-device
                    check(lib.mcrn_model_autotune(C.byref(self.d), self._ws.data_ptr(), nb,
                                                  torch.cuda.current_stream().cuda_stream), "mcrn_model_autotune")
                    # every replica runs rank 0's tiles: independent tuning could pick different ones (timing noise),
                    # i.e. different fp32 summation orders on different ranks
                    dp.share_autotune(self.group)
                self.out = torch.empty(B, To, N, od, device=dev)
                self.hatt, self.q, self.pos, self.neg = (torch.empty(B, N, D, device=dev) for _ in range(4))
            self.d_out = torch.empty(B, To, N, od, device=dev)
            self.d_q = torch.empty(B, N, D, device=dev)
            self._dims_key = key

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

    def _loss(self, out, labels, q, pos, neg, st):
        """3-term trainer loss and its gradients w.r.t. output and query: 3 HIP launches, no host sync."""
        d = self.d
        check(lib.mcrn_loss_fwd_bwd(d.B, d.T_out, d.N, d.output_dim, d.mem_dim, out.data_ptr(), labels.data_ptr(),
                                    q.data_ptr(), pos.data_ptr(), neg.data_ptr(), self.mean, self.std, self.lamb,
                                    self.lamb1, 1.0, self.loss_scratch.data_ptr(), self.losses.data_ptr(),
                                    self.d_out.data_ptr(), self.d_q.data_ptr(), st), "mcrn_loss_fwd_bwd")

    def train_step(self, x, ycov, labels):
        """One optimizer step; returns the (device) loss tensor without synchronising."""
        m = self.model
        x, ycov, labels = self._check_inputs(x, ycov, labels)
        self._prepare(x)
        st = torch.cuda.current_stream().cuda_stream
        if self.fused:
            teacher = m._teacher_flags(labels, self.batches_seen)
            tarr = (C.c_int * m.horizon)(*[int(f) for f in teacher])
            ps = Params(*[p.data_ptr() for p in self.params])
            gs = Grads(*[g.data_ptr() for g in self._gviews])
            nb = self._ws.numel()
            check(lib.mcrn_model_forward(C.byref(self.d), C.byref(ps), x.data_ptr(), ycov.data_ptr(), labels.data_ptr(),
                                         tarr, self._ws.data_ptr(), nb, self.out.data_ptr(), self.hatt.data_ptr(),
                                         self.q.data_ptr(), self.pos.data_ptr(), self.neg.data_ptr(), st),
                  "mcrn_model_forward")
            self._loss(self.out, labels, self.q, self.pos, self.neg, st)
            check(lib.mcrn_model_backward(C.byref(self.d), C.byref(ps), tarr, self.d_out.data_ptr(), None,
                                          self.d_q.data_ptr(), None, None, self._ws.data_ptr(), nb, C.byref(gs), st),
                  "mcrn_model_backward")
        else:
            # composed path (num_layers > 1): autograd over the per-cell HIP nodes, accumulating straight into the bucket
            self.flat_g.zero_()
            for p, g in zip(self.params, self._gviews):
                p.grad = g
            out, _hatt, q, pos, neg = m(x, ycov, labels, self.batches_seen)
            self._loss(out.detach().contiguous(), labels, q.detach().contiguous(), pos.detach().contiguous(),
                       neg.detach().contiguous(), st)
            torch.autograd.backward([out, q], [self.d_out, self.d_q])
            for p, g in zip(self.params, self._gviews):
                if p.grad is None or p.grad.data_ptr() != g.data_ptr():      # autograd replaced the tensor: copy it in
                    if p.grad is not None:
                        g.copy_(p.grad)
                    p.grad = g
        self.bucket.allreduce()                              # the single collective of the step (no-op at world 1)
        self.step_count += 1
        check(lib.mcrn_flat_clip_adam(self.flat_p.data_ptr(), self.flat_g.data_ptr(), self.m.data_ptr(),
                                      self.v.data_ptr(), self.n, self.lr, self.betas[0], self.betas[1], self.eps,
                                      self.step_count, self.max_grad_norm, self.bucket.grad_scale,
                                      self.scratch.data_ptr(), self.total_norm.data_ptr(), st), "mcrn_flat_clip_adam")
        self.batches_seen += 1
        # a fresh tensor for the caller (self.losses is rewritten by the next step) made by an element-wise kernel:
        # .clone() of a device scalar is a hipMemcpyAsync, ~100 us of queue idle per step on this runtime
        return self.losses[0].detach() * 1.0
