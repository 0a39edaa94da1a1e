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


def proto_mismatch_frac(got, want):
    """pos/neg are gathers Memory[top-k index]: a discrete choice.  Fraction of (b, n) rows whose prototype
    differs from the reference (near-ties between attention scores may legitimately resolve differently)."""
    got = np.asarray(got, np.float64).reshape(-1, got.shape[-1])
    want = np.asarray(want, np.float64).reshape(-1, want.shape[-1])
    bad = np.abs(got - want).max(axis=1) > 1e-6 * max(np.abs(want).max(), 1e-30)
    return float(bad.mean())
