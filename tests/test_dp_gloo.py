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


def _bucket_worker(rank, world, port, q):
    """Drives the PRODUCT bucket code (megacrn_amd.dp.FlatBucket, the class FlatTrainer steps through):
    flatten -> broadcast -> gradients written into the views -> ONE all-reduce -> grad_scale."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import megacrn_amd
    from megacrn_amd import dp
    from megacrn_amd._lib import PARAM_KEYS
    dp.init_from_env("gloo")
    rec, P, m = load_case("tiny", "f32")
    model = megacrn_amd.MegaCRN(num_nodes=m["N"], input_dim=1, output_dim=1, horizon=m["T_out"], rnn_units=m["H"],
                                mem_num=m["M"], mem_dim=m["D"])
    model.load_state_dict({k: torch.from_numpy(v + np.float32(rank)) for k, v in P.items()})   # ranks start DIFFERENT
    bucket = dp.FlatBucket(model._fused_params())
    after_bcast = {k: v.clone().numpy() for k, v in model.state_dict().items()}
    dp.seed_curriculum(99)
    teacher = O.curriculum_flags(m["T_out"], True, True, 15200, m["cl_decay"])
    B = 2                                                     # global batch of two samples, one per rank
    lo, hi = dp.shard_bounds(B, rank, world)
    G, _ = _shard_grad({k: v.astype(np.float64) for k, v in P.items()}, rec, m, lo, hi, teacher)
    for key, view in zip(PARAM_KEYS, bucket.grad_views):      # what mcrn_model_backward does on the GPU
        view.copy_(torch.from_numpy(G[key].astype(np.float32)))
    bucket.allreduce()
    q.put((rank, teacher, bucket.offsets, bucket.n, bucket.grad_scale, bucket.flat_g.numpy().copy(), after_bcast,
           [v.data_ptr() == bucket.flat_p[o:o + 1].data_ptr() for v, o in zip(model._fused_params(), bucket.offsets)]))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_flat_bucket_allreduce_and_grad_scale_world2():
    from megacrn_amd._lib import PARAM_KEYS
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bucket_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    rec, P, m = load_case("tiny", "f32")
    (_, teacher, offsets, n, scale, g0, sd0, is_view), (_, teacher1, _, _, _, g1, sd1, _) = res
    assert teacher == teacher1 and scale == 0.5 and all(is_view)
    assert all(o % 64 == 0 for o in offsets) and n % 64 == 0
    np.testing.assert_array_equal(g0, g1)                       # one all-reduce leaves identical buckets
    for k in P:                                                 # rank 0's weights won the broadcast, through the views
        np.testing.assert_array_equal(sd0[k], P[k]); np.testing.assert_array_equal(sd1[k], P[k])
    # the bucket, scaled by grad_scale, is the mean of the shard gradients = the oracle's DP gradient ...
    P64 = {k: v.astype(np.float64) for k, v in P.items()}
    shard = [_shard_grad(P64, rec, m, r, r + 1, teacher)[0] for r in range(world)]
    for key, off in zip(PARAM_KEYS, offsets):
        want = sum(g[key] for g in shard) / world
        got = g0[off:off + want.size].reshape(want.shape) * scale
        assert relerr(got, want) < 1e-6, key
    # ... and what clip + Adam then does with it is the single-process step on those mean gradients
    Gm = {k: sum(g[k] for g in shard) / world for k in P64}
    tot = O.clip_grad_norm(Gm, 5.0)
    flat = np.concatenate([(g0[o:o + P[k].size] * scale) for k, o in zip(PARAM_KEYS, offsets)])
    assert abs(np.sqrt((flat.astype(np.float64) ** 2).sum()) - tot) < 1e-5 * tot


# ---- tile-table sharing and bench.py's launch plumbing (no GPU call anywhere) ----------------------------------------
def _tune_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from megacrn_amd import dp, _lib
    dp.init_from_env("gloo")
    # ranks "tuned" differently: {kind 0 (tiled GEMM key, 9 words), cfg} and {kind 1 (bf16 GEMM key, 7 words: ..., role, nterm), cfg}
    mine = [0, 9, 1, 207, 4352, 207, 2, 1, 0, 1, 0, 3 + rank, 1, 7, 1, 7372, 1152, 1843, 1, 1, 1, 4 + rank]
    _lib.autotune_import(mine)
    assert _lib.autotune_export() == mine
    dp.share_autotune()
    q.put((rank, _lib.autotune_export()))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_ranks_adopt_rank0_tile_table_world2():
    """dp.share_autotune: after independent tuning every replica runs rank 0's tiles (product code over gloo; the table
    itself is host state of the library, no kernel runs)."""
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_tune_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0] == res[1] and res[0][11] == 3 and res[0][-1] == 4


def test_autotune_import_rejects_malformed_tables():
    from megacrn_amd import _lib
    # bf16 records: {1, 7 key words = btr, M, N, K, nsplit, role, nterm (a key word of its own since round 6), tile slot}
    # (last two: the retired bf16 tile slot 12 - round 2's stream-K tiles - and a record in round 5's 6-word key format)
    for bad in ([0, 9, 1], [2, 1, 0, 0], [0, 9] + [0] * 9 + [99], [1, 7] + [0] * 7 + [-1], [1, 7, 1, 7372, 1152, 1843, 1, 1, 1, 12],
                [1, 6, 1, 7372, 1152, 1843, 1, 1, 4]):
        with pytest.raises(RuntimeError):
            _lib.autotune_import(bad)
    _lib.lib.mcrn_autotune_clear()
    _lib.autotune_import([])                      # (import merges into the table: nothing to merge, nothing there)
    assert _lib.autotune_export() == []
    good = [1, 7, 1, 7372, 1152, 1843, 1, 1, 1, 4]
    _lib.autotune_import(good)
    _lib.autotune_import([1, 7, 1, 7372, 2048, 1843, 1, 1, 3, 9])       # a second shape (hi/lo pairs): merged, the first one stays
    _lib.autotune_import([1, 7, 1, 7372, 1024, 1843, 1, 1, 1, 10])      # slots 10 / 11: ping-pong tiles since round 6 (retired stream-K slots reused)
    assert len(_lib.autotune_export()) == 3 * len(good)
    _lib.lib.mcrn_autotune_clear()


def test_bench_self_launch_plumbing_gpus2(monkeypatch):
    """`python bench.py --gpus 2` without a torchrun environment: the parent must only build the torchrun command line
    (one rank per GPU, 127.0.0.1 rendezvous, its own arguments passed through) and hand it to a CHILD process - it must
    not touch the GPU itself.  subprocess.call is mocked: nothing is launched."""
    import subprocess
    import sys
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7

    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("HSA_ENABLE_IPC_MODE_LEGACY", raising=False)
    monkeypatch.setattr(torch.cuda, "is_available", lambda: (_ for _ in ()).throw(AssertionError("parent touched the GPU")))
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7                                    # the children's exit code is passed on
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=2" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    i = cmd.index(os.path.abspath(bench.__file__))
    assert cmd[i + 1:] == ["--gpus", "2", "--steps", "3", "--warmup", "1"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"     # dmabuf IPC: what RCCL needs on this pool
