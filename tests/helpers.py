"""Shared helpers for the parity tests (test infrastructure; may import oracle/)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SC_MEAN, SC_STD = 54.4, 19.5


def load_case(name, dn="f32"):
    z = np.load(os.path.join(GOLDEN, f"{name}_{dn}.npz"))
    rec = {k: z[k] for k in z.files}
    P = {k[2:]: v for k, v in rec.items() if k.startswith("p:")}
    meta = dict(zip(["B", "N", "T_in", "T_out", "H", "M", "D", "cheb_k", "num_layers", "cl_decay"],
                    [int(v) for v in rec["meta"]]))
    return rec, P, meta


def relerr(a, b):
    """max |a-b| / max|b|  (the 'rel' of north_star: relative to the tensor's scale)."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    den = max(np.abs(b).max(), 1e-30)
    return float(np.abs(a - b).max() / den)
