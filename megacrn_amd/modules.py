"""Drop-in module surface of the reference ``model/MegaCRN.py`` on top of libmegacrn_hip.so.

Same class names, constructor signatures, parameter names/shapes, ``state_dict`` keys and
5-tuple forward as the reference (``/root/reference/model/MegaCRN.py:7-194``), so that
``from MegaCRN import MegaCRN`` in ``model/traintest_MegaCRN.py:15`` can be pointed here
unchanged.  All device math runs in hand-written HIP kernels through the C ABI in
``include/megacrn_hip.h``; PyTorch supplies tensors, the autograd graph and the stream only.
There is no CPU path: tensors must live on a HIP device.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from ._lib import lib, check, Dims, Params, Grads, PARAM_FIELDS


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _need_gpu(t: torch.Tensor, who: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(f"{who}: megacrn_amd runs on MI355X only (got a {t.device} tensor); "
                           "there is no CPU fallback")


def _f32c(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        raise TypeError(f"megacrn_amd computes in fp32; got {t.dtype}")
    return t.contiguous()


def _p(t):
    return None if t is None else t.data_ptr()


class _WsPool:
    """Free-list of workspace buffers keyed by size.  A buffer that holds saved activations is
    owned by the autograd node until its backward has run."""

    def __init__(self):
        self.free = {}

    def take(self, nbytes: int, device) -> torch.Tensor:
        lst = self.free.get((nbytes, str(device)))
        if lst:
            return lst.pop()
        return torch.empty(nbytes, dtype=torch.uint8, device=device)

    def give(self, ws: torch.Tensor) -> None:
        self.free.setdefault((ws.numel(), str(ws.device)), []).append(ws)


_pool = _WsPool()


# --------------------------------------------------------------------------------------------
# autograd nodes
# --------------------------------------------------------------------------------------------
class _SupportsFn(torch.autograd.Function):
    """g1,g2 = softmax(relu(E1 E2^T)), softmax(relu(E2 E1^T))   (model/MegaCRN.py:169-172)."""

    @staticmethod
    def forward(ctx, We1, We2, Mem):
        _need_gpu(We1, "supports")
        We1, We2, Mem = _f32c(We1), _f32c(We2), _f32c(Mem)
        N, M = We1.shape
        D = Mem.shape[1]
        nb = lib.mcrn_supports_workspace_bytes(N, M, D)
        ws = _pool.take(nb, We1.device)
        g1 = torch.empty(N, N, device=We1.device)
        g2 = torch.empty(N, N, device=We1.device)
        check(lib.mcrn_supports_forward(N, M, D, _p(We1), _p(We2), _p(Mem), _p(ws), nb, _p(g1), _p(g2), _stream()),
              "mcrn_supports_forward")
        ctx.save_for_backward(We1, We2, Mem)
        ctx.ws = ws
        return g1, g2

    @staticmethod
    def backward(ctx, dg1, dg2):
        We1, We2, Mem = ctx.saved_tensors
        N, M = We1.shape
        D = Mem.shape[1]
        ws = ctx.ws
        dg1 = torch.zeros(N, N, device=We1.device) if dg1 is None else _f32c(dg1)
        dg2 = torch.zeros(N, N, device=We1.device) if dg2 is None else _f32c(dg2)
        dWe1, dWe2, dMem = torch.empty_like(We1), torch.empty_like(We2), torch.empty_like(Mem)
        check(lib.mcrn_supports_backward(N, M, D, _p(We1), _p(We2), _p(Mem), _p(dg1), _p(dg2), _p(ws), ws.numel(),
                                         _p(dWe1), _p(dWe2), _p(dMem), _stream()), "mcrn_supports_backward")
        _pool.give(ws)
        ctx.ws = None
        return dWe1, dWe2, dMem


class _AGCNFn(torch.autograd.Function):
    """AGCN.forward (model/MegaCRN.py:16-28)."""

    @staticmethod
    def forward(ctx, x, s1, s2, W, b, cheb_k):
        _need_gpu(x, "AGCN")
        x, s1, s2, W, b = map(_f32c, (x, s1, s2, W, b))
        B, N, Cc = x.shape
        O = W.shape[1]
        nb = lib.mcrn_agcn_workspace_bytes(B, N, Cc, O, cheb_k)
        if nb == 0:
            raise ValueError(f"AGCN: unsupported sizes / cheb_k={cheb_k} (supported: 2 .. 8)")
        ws = _pool.take(nb, x.device)
        y = torch.empty(B, N, O, device=x.device)
        check(lib.mcrn_agcn_forward(B, N, Cc, O, cheb_k, _p(x), _p(s1), _p(s2), _p(W), _p(b), _p(ws), nb, _p(y),
                                    _stream()), "mcrn_agcn_forward")
        ctx.save_for_backward(s1, s2, W)
        ctx.ws, ctx.dims = ws, (B, N, Cc, O, cheb_k)
        return y

    @staticmethod
    def backward(ctx, dy):
        s1, s2, W = ctx.saved_tensors
        B, N, Cc, O, K = ctx.dims
        ws = ctx.ws
        dy = _f32c(dy)
        dev = dy.device
        dx = torch.empty(B, N, Cc, device=dev)
        ds1, ds2 = torch.empty(N, N, device=dev), torch.empty(N, N, device=dev)
        dW, db = torch.empty_like(W), torch.empty(O, device=dev)
        check(lib.mcrn_agcn_backward(B, N, Cc, O, K, _p(dy), _p(s1), _p(s2), _p(W), _p(ws), ws.numel(), _p(dx),
                                     _p(ds1), _p(ds2), _p(dW), _p(db), _stream()), "mcrn_agcn_backward")
        _pool.give(ws)
        ctx.ws = None
        return dx, ds1, ds2, dW, db, None


class _CellFn(torch.autograd.Function):
    """AGCRNCell.forward (model/MegaCRN.py:38-48) with the GRU epilogues fused into the GEMMs."""

    @staticmethod
    def forward(ctx, x, h, s1, s2, gw, gb, uw, ub, cheb_k):
        _need_gpu(x, "AGCRNCell")
        x, h, s1, s2, gw, gb, uw, ub = map(_f32c, (x, h, s1, s2, gw, gb, uw, ub))
        B, N, din = x.shape
        H = h.shape[2]
        nb = lib.mcrn_cell_workspace_bytes(B, N, din, H, cheb_k)
        if nb == 0:
            raise ValueError(f"AGCRNCell: unsupported sizes / cheb_k={cheb_k} (supported: 2 .. 8)")
        ws = _pool.take(nb, x.device)
        hn = torch.empty(B, N, H, device=x.device)
        check(lib.mcrn_cell_forward(B, N, din, H, cheb_k, _p(x), _p(h), _p(s1), _p(s2), _p(gw), _p(gb), _p(uw),
                                    _p(ub), _p(ws), nb, _p(hn), _stream()), "mcrn_cell_forward")
        ctx.save_for_backward(s1, s2, gw, uw)
        ctx.ws, ctx.dims = ws, (B, N, din, H, cheb_k)
        return hn

    @staticmethod
    def backward(ctx, dhn):
        s1, s2, gw, uw = ctx.saved_tensors
        B, N, din, H, K = ctx.dims
        ws = ctx.ws
        dhn = _f32c(dhn)
        dev = dhn.device
        dx, dh = torch.empty(B, N, din, device=dev), torch.empty(B, N, H, device=dev)
        ds1, ds2 = torch.empty(N, N, device=dev), torch.empty(N, N, device=dev)
        dgw, duw = torch.empty_like(gw), torch.empty_like(uw)
        dgb, dub = torch.empty(2 * H, device=dev), torch.empty(H, device=dev)
        check(lib.mcrn_cell_backward(B, N, din, H, K, _p(dhn), _p(s1), _p(s2), _p(gw), _p(uw), _p(ws), ws.numel(),
                                     _p(dx), _p(dh), _p(ds1), _p(ds2), _p(dgw), _p(dgb), _p(duw), _p(dub),
                                     _stream()), "mcrn_cell_backward")
        _pool.give(ws)
        ctx.ws = None
        return dx, dh, ds1, ds2, dgw, dgb, duw, dub, None


class _MemoryFn(torch.autograd.Function):
    """MegaCRN.query_memory (model/MegaCRN.py:159-166)."""

    @staticmethod
    def forward(ctx, h, Mem, Wq):
        _need_gpu(h, "query_memory")
        h, Mem, Wq = map(_f32c, (h, Mem, Wq))
        B, N, H = h.shape
        M, D = Mem.shape
        nb = lib.mcrn_memory_workspace_bytes(B, N, H, M, D)
        ws = _pool.take(nb, h.device)
        dev = h.device
        val, q, pos, neg = (torch.empty(B, N, D, device=dev) for _ in range(4))
        ind = torch.empty(B, N, 2, dtype=torch.int32, device=dev)
        check(lib.mcrn_memory_forward(B, N, H, M, D, _p(h), _p(Mem), _p(Wq), _p(ws), nb, _p(val), _p(q), _p(pos),
                                      _p(neg), _p(ind), _stream()), "mcrn_memory_forward")
        ctx.save_for_backward(h, Mem, Wq)
        ctx.ws, ctx.dims = ws, (B, N, H, M, D)
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(ind)
        return val, q, pos, neg, ind

    @staticmethod
    def backward(ctx, dval, dq, dpos, dneg, _dind):
        h, Mem, Wq = ctx.saved_tensors
        B, N, H, M, D = ctx.dims
        ws = ctx.ws
        dev = h.device
        dval, dq, dpos, dneg = (None if t is None else _f32c(t) for t in (dval, dq, dpos, dneg))
        dh, dMem, dWq = torch.empty_like(h), torch.empty_like(Mem), torch.empty_like(Wq)
        check(lib.mcrn_memory_backward(B, N, H, M, D, _p(h), _p(Mem), _p(Wq), _p(dval), _p(dq), _p(dpos), _p(dneg),
                                       _p(ws), ws.numel(), _p(dh), _p(dMem), _p(dWq), _stream()),
              "mcrn_memory_backward")
        _pool.give(ws)
        ctx.ws = None
        return dh, dMem, dWq


class _ProjFn(torch.autograd.Function):
    """``proj`` = nn.Linear(H_dec, output_dim) of the multi-layer path (model/MegaCRN.py:144,186) on the library's own
    MFMA GEMM (``mcrn_gemm_f32``) - no vendor BLAS on any product path.  y = x W^T + b."""

    @staticmethod
    def forward(ctx, x, W, b):
        _need_gpu(x, "proj")
        x, W, b = _f32c(x), _f32c(W), _f32c(b)
        R, K, O = x.numel() // x.shape[-1], x.shape[-1], W.shape[0]
        y = b.expand(R, O).contiguous()
        check(lib.mcrn_gemm_f32(R, O, K, 0, 1, _p(x), _p(W), _p(y), 1.0, 1.0, 1, None, _stream()), "proj gemm")
        ctx.save_for_backward(x, W)
        return y.view(*x.shape[:-1], O)

    @staticmethod
    def backward(ctx, dy):
        x, W = ctx.saved_tensors
        dy = _f32c(dy)
        R, K, O = x.numel() // x.shape[-1], x.shape[-1], W.shape[0]
        dx, dW = torch.empty_like(x), torch.empty_like(W)
        st = _stream()
        check(lib.mcrn_gemm_f32(R, K, O, 0, 0, _p(dy), _p(W), _p(dx), 1.0, 0.0, 1, None, st), "proj dx")       # dy W
        slabs = torch.empty(64 * O * max(K, 1), device=x.device)
        check(lib.mcrn_gemm_f32(O, K, R, 1, 0, _p(dy), _p(x), _p(dW), 1.0, 0.0, 64, _p(slabs), st), "proj dW")  # dy^T x
        # bias gradient = dy^T 1, on the same GEMM (the ones vector is a fill, not arithmetic)
        db = torch.empty(O, device=x.device)
        ones = torch.ones(R, device=x.device)
        check(lib.mcrn_gemm_f32(O, 1, R, 1, 0, _p(dy), _p(ones), _p(db), 1.0, 0.0, 64, _p(slabs), st), "proj db")
        return dx, dW, db


class _ModelFn(torch.autograd.Function):
    """Whole MegaCRN.forward (model/MegaCRN.py:168-194) as one autograd node (num_layers == 1)."""

    @staticmethod
    def forward(ctx, dims, teacher, x, ycov, labels, *params):
        d: Dims = dims
        dev = x.device
        params = tuple(_f32c(p) for p in params)
        x, ycov = _f32c(x), _f32c(ycov)
        labels = None if labels is None else _f32c(labels)
        nb = lib.mcrn_model_workspace_bytes(C.byref(d))
        if nb == 0:
            raise ValueError("MegaCRN: " + lib.mcrn_last_error().decode())
        ws = _pool.take(nb, dev)
        B, N, To, od, D = d.B, d.N, d.T_out, d.output_dim, d.mem_dim
        out = torch.empty(B, To, N, od, device=dev)
        hatt, q, pos, neg = (torch.empty(B, N, D, device=dev) for _ in range(4))
        ps = Params(*[p.data_ptr() for p in params])
        tarr = (C.c_int * To)(*[int(bool(v)) for v in teacher]) if teacher is not None else None
        check(lib.mcrn_model_forward(C.byref(d), C.byref(ps), _p(x), _p(ycov), _p(labels), tarr, _p(ws), nb,
                                     _p(out), _p(hatt), _p(q), _p(pos), _p(neg), _stream()), "mcrn_model_forward")
        ctx.set_materialize_grads(False)
        if any(ctx.needs_input_grad):      # grad mode is off inside forward(); this is the outer view
            ctx.save_for_backward(*params)
            ctx.ws, ctx.d, ctx.tarr = ws, d, tarr
        else:
            _pool.give(ws)
        return out, hatt, q, pos, neg

    @staticmethod
    def backward(ctx, d_out, d_hatt, d_q, d_pos, d_neg):
        params = ctx.saved_tensors
        d, ws = ctx.d, ctx.ws
        if ws is None:
            raise RuntimeError("MegaCRN backward called twice (saved activations were released)")
        dev = params[0].device
        if d_out is None:
            d_out = torch.zeros(d.B, d.T_out, d.N, d.output_dim, device=dev)
        d_out, d_hatt, d_q, d_pos, d_neg = (None if t is None else _f32c(t) for t in (d_out, d_hatt, d_q, d_pos, d_neg))
        grads = tuple(torch.empty_like(p) for p in params)
        ps = Params(*[p.data_ptr() for p in params])
        gs = Grads(*[g.data_ptr() for g in grads])
        check(lib.mcrn_model_backward(C.byref(d), C.byref(ps), ctx.tarr, _p(d_out), _p(d_hatt), _p(d_q), _p(d_pos),
                                      _p(d_neg), _p(ws), ws.numel(), C.byref(gs), _stream()), "mcrn_model_backward")
        _pool.give(ws)
        ctx.ws = None
        return (None, None, None, None, None) + grads


# --------------------------------------------------------------------------------------------
# module surface (names, ctor signatures and parameter shapes of model/MegaCRN.py)
# --------------------------------------------------------------------------------------------
class AGCN(nn.Module):
    def __init__(self, dim_in, dim_out, cheb_k):
        super().__init__()
        self.cheb_k = cheb_k
        self.weights = nn.Parameter(torch.empty(2 * cheb_k * dim_in, dim_out))   # 2 = number of supports
        self.bias = nn.Parameter(torch.empty(dim_out))
        nn.init.xavier_normal_(self.weights)
        nn.init.constant_(self.bias, val=0)

    def forward(self, x, supports):
        if len(supports) != 2:
            raise ValueError("AGCN expects exactly two supports (weights are sized 2*cheb_k*dim_in)")
        return _AGCNFn.apply(x, supports[0], supports[1], self.weights, self.bias, self.cheb_k)


class AGCRNCell(nn.Module):
    def __init__(self, node_num, dim_in, dim_out, cheb_k):
        super().__init__()
        self.node_num = node_num
        self.hidden_dim = dim_out
        self.gate = AGCN(dim_in + self.hidden_dim, 2 * dim_out, cheb_k)
        self.update = AGCN(dim_in + self.hidden_dim, dim_out, cheb_k)

    def forward(self, x, state, supports):
        # x: (B, num_nodes, input_dim), state: (B, num_nodes, hidden_dim)
        state = state.to(x.device)
        return _CellFn.apply(x, state, supports[0], supports[1], self.gate.weights, self.gate.bias,
                             self.update.weights, self.update.bias, self.gate.cheb_k)

    def init_hidden_state(self, batch_size):
        return torch.zeros(batch_size, self.node_num, self.hidden_dim)


class ADCRNN_Encoder(nn.Module):
    def __init__(self, node_num, dim_in, dim_out, cheb_k, num_layers):
        super().__init__()
        assert num_layers >= 1, 'At least one DCRNN layer in the Encoder.'
        self.node_num = node_num
        self.input_dim = dim_in
        self.num_layers = num_layers
        self.dcrnn_cells = nn.ModuleList()
        self.dcrnn_cells.append(AGCRNCell(node_num, dim_in, dim_out, cheb_k))
        for _ in range(1, num_layers):
            self.dcrnn_cells.append(AGCRNCell(node_num, dim_out, dim_out, cheb_k))

    def forward(self, x, init_state, supports):
        """x (B,T,N,dim_in), init_state: one (B,N,hidden) per layer -> (states of the last layer
        (B,T,N,hidden), [last state of every layer])   (reference :65-83)."""
        assert x.shape[2] == self.node_num and x.shape[3] == self.input_dim
        layer_in, last_states = x, []
        for cell, state in zip(self.dcrnn_cells, init_state):
            steps = []
            for x_t in layer_in.unbind(dim=1):
                state = cell(x_t, state, supports)
                steps.append(state)
            last_states.append(state)
            layer_in = torch.stack(steps, dim=1)
        return layer_in, last_states

    def init_hidden(self, batch_size):
        return [cell.init_hidden_state(batch_size) for cell in self.dcrnn_cells]


class ADCRNN_Decoder(nn.Module):
    def __init__(self, node_num, dim_in, dim_out, cheb_k, num_layers):
        super().__init__()
        assert num_layers >= 1, 'At least one DCRNN layer in the Decoder.'
        self.node_num = node_num
        self.input_dim = dim_in
        self.num_layers = num_layers
        self.dcrnn_cells = nn.ModuleList()
        self.dcrnn_cells.append(AGCRNCell(node_num, dim_in, dim_out, cheb_k))
        for _ in range(1, num_layers):
            self.dcrnn_cells.append(AGCRNCell(node_num, dim_out, dim_out, cheb_k))

    def forward(self, xt, init_state, supports):
        """One decoder step through the layer stack: xt (B,N,dim_in) -> (top state, [state per layer])
        (reference :103-113)."""
        assert xt.shape[1] == self.node_num and xt.shape[2] == self.input_dim
        states, inp = [], xt
        for cell, prev in zip(self.dcrnn_cells, init_state):
            inp = cell(inp, prev, supports)
            states.append(inp)
        return inp, states


class MegaCRN(nn.Module):
    def __init__(self, num_nodes, input_dim, output_dim, horizon, rnn_units, num_layers=1, cheb_k=3,
                 ycov_dim=1, mem_num=20, mem_dim=64, cl_decay_steps=2000, use_curriculum_learning=True):
        super().__init__()
        self.num_nodes = num_nodes
        self.input_dim = input_dim
        self.rnn_units = rnn_units
        self.output_dim = output_dim
        self.horizon = horizon
        self.num_layers = num_layers
        self.cheb_k = cheb_k
        self.ycov_dim = ycov_dim
        self.cl_decay_steps = cl_decay_steps
        self.use_curriculum_learning = use_curriculum_learning
        self.precision = _lib.default_precision()   # _lib.F32 (exact) or _lib.BF16X3

        # memory
        self.mem_num = mem_num
        self.mem_dim = mem_dim
        self.memory = self.construct_memory()

        # encoder
        self.encoder = ADCRNN_Encoder(self.num_nodes, self.input_dim, self.rnn_units, self.cheb_k, self.num_layers)

        # decoder
        self.decoder_dim = self.rnn_units + self.mem_dim
        self.decoder = ADCRNN_Decoder(self.num_nodes, self.output_dim + self.ycov_dim, self.decoder_dim,
                                      self.cheb_k, self.num_layers)

        # output
        self.proj = nn.Sequential(nn.Linear(self.decoder_dim, self.output_dim, bias=True))

    def compute_sampling_threshold(self, batches_seen):
        return self.cl_decay_steps / (self.cl_decay_steps + np.exp(batches_seen / self.cl_decay_steps))

    def construct_memory(self):
        memory_dict = nn.ParameterDict()
        memory_dict['Memory'] = nn.Parameter(torch.randn(self.mem_num, self.mem_dim), requires_grad=True)
        memory_dict['Wq'] = nn.Parameter(torch.randn(self.rnn_units, self.mem_dim), requires_grad=True)
        memory_dict['We1'] = nn.Parameter(torch.randn(self.num_nodes, self.mem_num), requires_grad=True)
        memory_dict['We2'] = nn.Parameter(torch.randn(self.num_nodes, self.mem_num), requires_grad=True)
        for param in memory_dict.values():
            nn.init.xavier_normal_(param)
        return memory_dict

    def query_memory(self, h_t: torch.Tensor):
        value, query, pos, neg, _ = _MemoryFn.apply(h_t, self.memory['Memory'], self.memory['Wq'])
        return value, query, pos, neg

    # -- curriculum decisions: one numpy draw per decoder step, only in training (reference :188-190)
    def _teacher_flags(self, labels, batches_seen):
        flags = []
        for _ in range(self.horizon):
            f = False
            if self.training and self.use_curriculum_learning:
                c = np.random.uniform(0, 1)
                f = bool(c < self.compute_sampling_threshold(batches_seen))
            flags.append(f)
        return flags

    def _check_inputs(self, x, y_cov, labels):
        """The fused C entry point takes raw pointers: reject here what the reference rejects with a shape
        error in torch.cat / torch.stack (model/MegaCRN.py:67,106,185,192)."""
        if x.dim() != 4 or x.shape[2] != self.num_nodes or x.shape[3] != self.input_dim:
            raise ValueError(f"x must be (B, T_in, {self.num_nodes}, {self.input_dim}); got {tuple(x.shape)}")
        B = x.shape[0]
        want_c = (B, self.horizon, self.num_nodes, self.ycov_dim)
        if y_cov is None or tuple(y_cov.shape[:3]) != want_c[:3] or y_cov.shape[3] < self.ycov_dim:
            raise ValueError(f"y_cov must be {want_c}; got {None if y_cov is None else tuple(y_cov.shape)}")
        if y_cov.shape[3] != self.ycov_dim:
            raise ValueError(f"y_cov must be {want_c}; got {tuple(y_cov.shape)}")
        if labels is not None and tuple(labels.shape) != (B, self.horizon, self.num_nodes, self.output_dim):
            raise ValueError(f"labels must be {(B, self.horizon, self.num_nodes, self.output_dim)}; got {tuple(labels.shape)}")
        for name, t in (("y_cov", y_cov), ("labels", labels)):
            if t is not None and t.device != x.device:
                raise ValueError(f"{name} lives on {t.device}, x on {x.device}")

    def _fused_params(self):
        e, dcd = self.encoder.dcrnn_cells[0], self.decoder.dcrnn_cells[0]
        return (self.memory['Memory'], self.memory['Wq'], self.memory['We1'], self.memory['We2'],
                e.gate.weights, e.gate.bias, e.update.weights, e.update.bias,
                dcd.gate.weights, dcd.gate.bias, dcd.update.weights, dcd.update.bias,
                self.proj[0].weight, self.proj[0].bias)

    def forward(self, x, y_cov, labels=None, batches_seen=None):
        _need_gpu(x, "MegaCRN.forward")
        if self.num_layers == 1:
            teacher = self._teacher_flags(labels, batches_seen)
            if any(teacher) and labels is None:
                raise ValueError("curriculum learning needs labels")
            d = Dims(x.shape[0], self.num_nodes, x.shape[1], self.horizon, self.input_dim, self.output_dim,
                     self.ycov_dim, self.rnn_units, self.mem_num, self.mem_dim, self.cheb_k, self.precision)
            self._check_inputs(x, y_cov, labels)
            return _ModelFn.apply(d, teacher, x, y_cov, labels, *self._fused_params())
        return self._forward_composed(x, y_cov, labels, batches_seen)

    def _forward_composed(self, x, y_cov, labels=None, batches_seen=None):
        """num_layers > 1: same data flow as the reference forward (:168-194), every arithmetic op a HIP node of this
        library (supports, cells, memory head, projection); torch only concatenates and stacks tensors."""
        B = x.shape[0]
        supports = list(_SupportsFn.apply(self.memory['We1'], self.memory['We2'], self.memory['Memory']))
        zeros = [s.to(x.device) for s in self.encoder.init_hidden(B)]
        enc_states, _ = self.encoder(x, zeros, supports)
        h_last = enc_states[:, -1]
        h_att, query, pos, neg = self.query_memory(h_last)
        dec_state = [torch.cat([h_last, h_att], dim=-1)] * self.num_layers
        teacher = self._teacher_flags(labels, batches_seen)
        go = x.new_zeros(B, self.num_nodes, self.output_dim)
        preds = []
        for t in range(self.horizon):
            top, dec_state = self.decoder(torch.cat([go, y_cov[:, t]], dim=-1), dec_state, supports)
            go = _ProjFn.apply(top, self.proj[0].weight, self.proj[0].bias)
            preds.append(go)
            if teacher[t]:
                go = labels[:, t]
        return torch.stack(preds, dim=1), h_att, query, pos, neg


def print_params(model):
    param_count = 0
    print('Trainable parameter list:')
    for name, param in model.named_parameters():
        if param.requires_grad:
            print(name, param.shape, param.numel())
            param_count += param.numel()
    print(f'In total: {param_count} trainable parameters. \n')
    return
