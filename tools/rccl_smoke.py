#!/usr/bin/env python3
"""Hardware check of the collective path with the ranks this box has (1 GPU -> world_size 1 over RCCL):
process-group init on the device, broadcast of the flat parameter bucket, all-reduce of the flat gradient bucket
between backward and the fused clip+Adam, on the stream order FlatTrainer uses.  With 1 rank the collectives are
identities, so the loss trajectory must equal a non-distributed run bit for bit.
    python tools/rccl_smoke.py            (or under torch.distributed.run with more ranks on a multi-GPU node)"""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import megacrn_amd  # noqa: E402
from megacrn_amd import dp  # noqa: E402
from megacrn_amd.trainer import FlatTrainer  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
rank = int(os.environ.get("RANK", "0"))
world = int(os.environ.get("WORLD_SIZE", "1"))
lr = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(lr)
dev = torch.device("cuda", lr)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
cfg = dict(bench.CONFIGS["metrla"]); cfg["B"] = 8


def run(distributed):
    torch.manual_seed(7)
    model = megacrn_amd.MegaCRN(cfg["N"], 1, 1, cfg["T"], cfg["H"], mem_num=cfg["M"], mem_dim=cfg["D"]).to(dev).train()
    tr = FlatTrainer(model, scaler_mean=54.4, scaler_std=19.5)
    x, yc, y = bench.synth(cfg, cfg["B"], 11 + rank, dev)
    losses = []
    for _ in range(4):
        losses.append(float(tr.train_step(x, yc, y)))
        if distributed:
            # the collectives FlatTrainer issues when world > 1, forced here on the live buckets for world == 1 too
            dp.allreduce_flat(tr.flat_g, None)
            dp.broadcast_flat(tr.flat_p, None)
            torch.cuda.synchronize()
    return losses


a = run(False)
b = run(True)
t = torch.tensor([1.0 + rank], device=dev)
dist.all_reduce(t)
if rank == 0:
    print("losses (plain)      :", [round(v, 6) for v in a])
    print("losses (collectives):", [round(v, 6) for v in b])
    print("all_reduce sum of (1+rank) over", world, "ranks =", float(t), " backend", dist.get_backend())
    assert float(t) == sum(1.0 + r for r in range(world))
    if world == 1:
        assert a == b, "1-rank collectives must be identities"
    print("rccl smoke ok")
dist.destroy_process_group()
