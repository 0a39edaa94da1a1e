"""world_size-2 data-parallel checks on CPU (gloo): the DP plumbing of megacrn_amd.dp - batch
sharding, the single flat-bucket all-reduce, 1/world scaling, shared curriculum stream - reproduces
the single-process gradient.  Per-rank gradients come from the oracle (test infrastructure); the
collective path is the product code."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import load_case, relerr, SC_MEAN, SC_STD
from oracle import megacrn_oracle as O


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _flat(G, keys):
    return np.concatenate([G[k].reshape(-1) for k in keys])


def _shard_grad(P, rec, m, lo, hi, teacher):
    x, yc, y = rec["x"][lo:hi], rec["ycov"][lo:hi], rec["labels"][lo:hi]
    outs, cache = O.model_fwd(P, x, yc, y, teacher, cheb_k=m["cheb_k"], num_layers=m["num_layers"])
    losses, d_out, d_q = O.loss_fwd_bwd(outs, y, SC_MEAN, SC_STD)
    G, _ = O.model_bwd(d_out, cache, d_query=d_q)
    return G, losses[0]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from megacrn_amd import dp
    rk, _, ws = dp.init_from_env("gloo")
    assert (rk, ws) == (rank, world) and dp.world_size() == world and dp.rank() == rank
    rec, P, m = load_case("tiny", "f64")
    keys = list(P.keys())
    B = rec["x"].shape[0]
    # every rank draws the SAME curriculum flags from the shared numpy stream (model/MegaCRN.py:189)
    dp.seed_curriculum(1234)
    teacher = O.curriculum_flags(m["T_out"], True, True, 15200, m["cl_decay"])
    lo, hi = dp.shard_bounds(B, rank, world)
    G, _ = _shard_grad(P, rec, m, lo, hi, teacher)
    flat = torch.from_numpy(_flat(G, keys).copy())
    dp.allreduce_flat(flat)                       # the one collective of the step
    flat /= world                                 # (done inside mcrn_flat_clip_adam on the GPU path)
    # parameters broadcast from rank 0 leave every rank identical
    pf = torch.from_numpy(_flat(P, keys).copy()) + rank
    dp.broadcast_flat(pf)
    q.put((rank, lo, hi, teacher, flat.numpy(), pf.numpy()))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_dp_allreduce_matches_mean_of_shard_gradients():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    rec, P, m = load_case("tiny", "f64")
    keys = list(P.keys())
    B = rec["x"].shape[0]
    # shards partition the batch
    assert res[0][1] == 0 and res[-1][2] == B and res[0][2] == res[1][1]
    assert res[0][3] == res[1][3], "curriculum flags must be identical on all ranks"
    np.testing.assert_array_equal(res[0][4], res[1][4])      # all-reduce leaves identical buffers
    np.testing.assert_array_equal(res[0][5], res[1][5])
    np.testing.assert_array_equal(res[0][5], _flat(P, keys))  # rank 0's parameters won
    # reference value: mean over ranks of the per-shard gradients (DDP semantics: each rank's
    # masked-MAE is normalised by its local mask mean, model/utils.py:127-128)
    teacher = res[0][3]
    want = sum(_flat(_shard_grad(P, rec, m, lo, hi, teacher)[0], keys) for _, lo, hi, *_ in res) / world
    assert relerr(res[0][4], want) < 1e-12


def test_shard_bounds_cover_batch():
    from megacrn_amd import dp
    for B in (1, 7, 64, 65):
        for w in (1, 2, 3, 8):
            spans = [dp.shard_bounds(B, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_equal_mask_fraction_makes_dp_exact():
    """With equal mask fractions per shard the DP mean equals the single-process full-batch gradient
    (SURVEY.md 8(e)): triplet/MSE terms are plain means, masked-MAE normalisers coincide."""
    rec, P, m = load_case("tiny", "f64")
    rec = dict(rec)
    y = rec["labels"].copy()
    y[y == (0.0 - SC_MEAN) / SC_STD] = 0.3        # remove masked entries -> mask fraction 1 everywhere
    rec["labels"] = y
    keys = list(P.keys())
    teacher = [False, True, False, True]
    B = y.shape[0]
    # B=3 does not split evenly; use the first 2 samples as the "global batch"
    rec2 = {k: (v[:2] if k in ("x", "ycov", "labels") else v) for k, v in rec.items()}
    full, _ = _shard_grad(P, rec2, m, 0, 2, teacher)
    parts = [_shard_grad(P, rec2, m, i, i + 1, teacher)[0] for i in range(2)]
    mean = sum(_flat(g, keys) for g in parts) / 2
    assert relerr(mean, _flat(full, keys)) < 1e-10
