"""debug: METR-LA-shaped eval forward through the streaming weight pool, enc/dec selectable (MCRN_WP_DBG)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import megacrn_amd
from oracle import megacrn_oracle as O
N, B, T, H = int(os.environ.get("N", 207)), int(os.environ.get("B", 2)), int(os.environ.get("T", 2)), 64
P = O.init_params(N, rnn_units=H, seed=3)
rng = np.random.default_rng(9)
x = rng.standard_normal((B, T, N, 1)).astype(np.float32)
ycov = rng.random((B, T, N, 1)).astype(np.float32)
model = megacrn_amd.MegaCRN(N, 1, 1, T, H)
model.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()})
model = model.cuda().eval()
if os.environ.get("TRAIN"):
    o = model(torch.from_numpy(x).cuda(), torch.from_numpy(ycov).cuda())
    torch.cuda.synchronize(); print("fwd done", flush=True)
    (o[0].sum() + o[2].sum()).backward()
    torch.cuda.synchronize(); print("bwd done", flush=True)
else:
    with torch.no_grad():
        o = model(torch.from_numpy(x).cuda(), torch.from_numpy(ycov).cuda())
    torch.cuda.synchronize()
o = [t.detach() for t in o]
ref, _ = O.model_fwd(P, x, ycov)
print("ok", float(np.abs(o[0].cpu().numpy() - ref[0]).max() / np.abs(ref[0]).max()))
