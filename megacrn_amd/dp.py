"""Batch data parallelism for the MegaCRN hot path: one process per GPU, samples sharded across
ranks, parameters replicated, ONE all-reduce(sum) of the flat gradient bucket per optimizer step
(0.5-2.8 MB, latency-bound on xGMI - SURVEY.md 5.8), then 1/world scaling inside the fused
clip+Adam kernel.  No other exchange; the batch-independent adjacency is recomputed on every rank.
Works with backend "nccl" (= RCCL on ROCm) on GPUs and "gloo" on CPU (tests)."""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.distributed as dist


def world_size(group=None) -> int:
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def rank(group=None) -> int:
    return dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0


def init_from_env(backend: str | None = None) -> tuple[int, int, int]:
    """torchrun-style env (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR/PORT) -> (rank, local_rank, world)."""
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    rk = int(os.environ.get("RANK", "0"))
    lr = int(os.environ.get("LOCAL_RANK", "0"))
    if ws > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(lr)
            dist.init_process_group(backend, rank=rk, world_size=ws, device_id=torch.device("cuda", lr))
        else:
            dist.init_process_group(backend, rank=rk, world_size=ws)
    return rk, lr, ws


def shard_bounds(global_batch: int, rk: int, ws: int) -> tuple[int, int]:
    """rank r owns the r-th block of consecutive samples (sizes differ by at most one).  The step averages per-rank
    gradients with weight 1/world, so it equals the global-batch gradient only for EQUAL shards (and equal mask
    fractions, model/utils.py:127-128); megacrn_amd.train requires batch_size % world == 0."""
    base, rem = divmod(global_batch, ws)
    lo = rk * base + min(rk, rem)
    return lo, lo + base + (1 if rk < rem else 0)


def allreduce_flat(flat_g: torch.Tensor, group=None) -> None:
    dist.all_reduce(flat_g, op=dist.ReduceOp.SUM, group=group)


def broadcast_flat(flat_p: torch.Tensor, group=None, src: int = 0) -> None:
    dist.broadcast(flat_p, src=src, group=group)


def share_autotune(group=None, src: int = 0) -> None:
    """Every rank adopts rank `src`'s GEMM tile table (mcrn_model_autotune times tiles on-device; noise could choose
    different tiles - different fp32 summation orders - on different replicas).  No-op at world size 1."""
    if world_size(group) == 1:
        return
    from . import _lib
    box = [_lib.autotune_export() if rank(group) == src else None]
    dist.broadcast_object_list(box, src=src, group=group)
    if rank(group) != src:
        _lib.autotune_import(box[0])


def seed_curriculum(seed: int) -> None:
    """The curriculum draw is one np.random.uniform() per decoder step for the whole batch
    (model/MegaCRN.py:189): every rank must consume the same stream, so all ranks use `seed`."""
    np.random.seed(seed)


class FlatBucket:
    """The step's single message: every parameter lives in ONE flat fp32 buffer (`flat_p`), every gradient in a
    second one (`flat_g`), each tensor in a 64-float (256 B) aligned slice.  `nn.Parameter`s of the reference shapes
    are re-pointed at views of `flat_p` (state_dict keeps working), the backward pass writes gradients straight into
    the views of `flat_g`, `allreduce()` is the ONE collective of a step, and `grad_scale` (= 1/world) is applied by
    the consumer of the bucket (the fused clip+Adam kernel).  Device-agnostic: the CPU tests drive exactly this class
    over gloo, the GPU path over RCCL."""

    def __init__(self, params, group=None):
        self.params = list(params)
        self.group = group
        self.world = world_size(group)
        if not self.params:
            raise ValueError("FlatBucket: no parameters")
        dev = self.params[0].device
        self.sizes = [p.numel() for p in self.params]
        self.offsets, o = [], 0
        for n in self.sizes:
            self.offsets.append(o)
            o += (n + 63) // 64 * 64
        self.n = o
        self.flat_p = torch.zeros(o, device=dev)
        self.flat_g = torch.zeros(o, device=dev)
        self.grad_views = []
        for p, off, n in zip(self.params, self.offsets, self.sizes):
            if p.dtype != torch.float32 or p.device != dev:
                raise ValueError("FlatBucket: parameters must be fp32 tensors on one device")
            self.flat_p[off:off + n].copy_(p.detach().reshape(-1))
            p.data = self.flat_p[off:off + n].view(p.shape)
            self.grad_views.append(self.flat_g[off:off + n].view(p.shape))
        if self.world > 1:
            broadcast_flat(self.flat_p, group)                 # identical weights on every rank

    @property
    def grad_scale(self) -> float:
        return 1.0 / self.world

    def allreduce(self) -> None:
        if self.world > 1:
            allreduce_flat(self.flat_g, self.group)
