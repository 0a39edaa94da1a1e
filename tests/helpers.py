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


def relerr_rows(a, b, floor=1e-3):
    """Per-row normalised error of a 2-D gradient: max_r ( max|a_r - b_r| / max(max|b_r|, floor * max|b|) ).
    `relerr` is relative to the whole tensor's largest entry, so a small-magnitude block (one row of dWe1, one
    input-channel block of an AGCN weight) could be far off and pass; here every row is held to its own scale,
    with rows below `floor` of the global scale normalised by that floor (their absolute error still counts)."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    if a.ndim == 1:
        a, b = a[None], b[None]
    a, b = a.reshape(a.shape[0], -1), b.reshape(b.shape[0], -1)
    g = max(np.abs(b).max(), 1e-30)
    den = np.maximum(np.abs(b).max(axis=1), floor * g)
    return float((np.abs(a - b).max(axis=1) / den).max())


def proto_gap_check(got_pos, got_neg, cmem, gap=1e-5):
    """pos / neg = Memory[top-2 indices] is INDEX work (model/MegaCRN.py:163-165): the prototypes must be the oracle's wherever
    the oracle's choice is decided by more than the arithmetic tolerance.  `cmem` is the oracle's memory-head cache
    (h, Mem, Wq, q, att, ind).  A row is EXCUSED only when the score gap that decides it - top-1 vs top-2 for pos, and either
    that or top-2 vs top-3 for neg - is below `gap` x max(1, |score|max of the row) (two fp32 evaluations of the same
    scores differ by that much); every other row must match bit-exactly in its index, i.e. the gathered prototype rows
    must be identical.  Returns (unexcused mismatches, excused rows, total rows, smallest deciding gap / tolerance among
    the mismatched rows) - the caller asserts the first is 0 and reports the second."""
    _h, Mem, _Wq, q, _att, ind = cmem
    Mem = np.asarray(Mem, np.float64)
    sc = np.asarray(q, np.float64).reshape(-1, Mem.shape[1]) @ Mem.T
    srt = -np.sort(-sc, axis=1)
    tol = gap * np.maximum(1.0, np.abs(sc).max(axis=1))
    g12 = srt[:, 0] - srt[:, 1]
    g23 = srt[:, 1] - srt[:, 2] if sc.shape[1] > 2 else np.full_like(g12, np.inf)
    ind = np.asarray(ind).reshape(-1, 2)
    want_pos, want_neg = Mem[ind[:, 0]], Mem[ind[:, 1]]
    D = Mem.shape[1]
    bad_pos = np.abs(np.asarray(got_pos, np.float64).reshape(-1, D) - want_pos).max(axis=1) > 0
    bad_neg = np.abs(np.asarray(got_neg, np.float64).reshape(-1, D) - want_neg).max(axis=1) > 0
    exc_pos, exc_neg = g12 <= tol, np.minimum(g12, g23) <= tol
    unexcused = (bad_pos & ~exc_pos) | (bad_neg & ~exc_neg)
    worst = np.inf
    if (bad_pos | bad_neg).any():
        worst = float(np.minimum(np.where(bad_pos, g12 / tol, np.inf), np.where(bad_neg, np.minimum(g12, g23) / tol, np.inf)).min())
    return int(unexcused.sum()), int((exc_pos | exc_neg).sum()), int(len(g12)), worst
