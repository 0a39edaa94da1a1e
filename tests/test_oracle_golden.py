"""Pins oracle/megacrn_oracle.py against vectors produced by the reference itself
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

from oracle import megacrn_oracle as O
from helpers import load_case, relerr, SC_MEAN, SC_STD

CASES = [("tiny", "f32"), ("tiny", "f64"), ("odd", "f32"), ("odd", "f64"),
         ("layers2", "f32"), ("layers2", "f64"), ("cheb2", "f32"), ("cheb2", "f64"),
         ("metrla", "f32"), ("cheb4", "f32"), ("cheb4", "f64")]
TOL = {"f32": 2e-5, "f64": 1e-11}
GTOL = {"f32": 2e-4, "f64": 1e-10}   # gradients: fp32 accumulation-order noise through 2T cells


@pytest.mark.parametrize("name,dn", CASES)
def test_forward_eval(name, dn):
    rec, P, m = load_case(name, dn)
    outs, cache = O.model_fwd(P, rec["x"], rec["ycov"], cheb_k=m["cheb_k"], num_layers=m["num_layers"])
    assert relerr(cache["sup"][0], rec["g1"]) < TOL[dn]
    assert relerr(cache["sup"][1], rec["g2"]) < TOL[dn]
    for nm, o in zip(("output", "h_att", "query", "pos", "neg"), outs):
        assert o.dtype == rec["x"].dtype
        assert relerr(o, rec["eval:" + nm]) < TOL[dn], nm


@pytest.mark.parametrize("name,dn", CASES)
def test_train_step(name, dn):
    rec, P, m = load_case(name, dn)
    teacher = [bool(v) for v in rec["teacher"]]
    outs, cache = O.model_fwd(P, rec["x"], rec["ycov"], rec["labels"], teacher,
                              cheb_k=m["cheb_k"], num_layers=m["num_layers"])
    for nm, o in zip(("output", "h_att", "query", "pos", "neg"), outs):
        assert relerr(o, rec["train:" + nm]) < TOL[dn], nm
    losses, d_out, d_q = O.loss_fwd_bwd(outs, rec["labels"], SC_MEAN, SC_STD)
    np.testing.assert_allclose(losses, rec["train:loss"], rtol=10 * TOL[dn])
    G, _ = O.model_bwd(d_out, cache, d_query=d_q)
    for k, g in G.items():
        assert relerr(g, rec["g:" + k]) < GTOL[dn], k
    gn = O.clip_grad_norm(G, 5.0)
    assert abs(gn - float(rec["train:gnorm"])) / float(rec["train:gnorm"]) < GTOL[dn]
    if any(k.startswith("p1:") for k in rec):
        P1 = {k: v.copy() for k, v in P.items()}
        opt = O.Adam(P1, lr=0.01, eps=1e-3)
        opt.step(P1, G)
        for k in P1:
            assert relerr(P1[k], rec["p1:" + k]) < 10 * GTOL[dn], k


@pytest.mark.parametrize("name,dn", [("tiny", "f64"), ("odd", "f64"), ("layers2", "f32")])
def test_loss_trajectory(name, dn):
    rec, P, m = load_case(name, dn)
    P = {k: v.copy() for k, v in P.items()}
    opt = O.Adam(P, lr=0.01, eps=1e-3)
    got = []
    for s in range(3):
        teacher = [bool(v) for v in rec["traj:teacher"][s]]
        losses, _, _ = O.train_step(P, opt, rec["x"], rec["ycov"], rec["labels"], teacher,
                                    SC_MEAN, SC_STD, cheb_k=m["cheb_k"], num_layers=m["num_layers"])
        got.append(losses[0])
    np.testing.assert_allclose(got, rec["traj:loss"], rtol=1e-4 if dn == "f32" else 1e-9)


@pytest.mark.parametrize("dn", ["f32", "f64"])
def test_ops(dn, golden_dir):
    z = np.load(f"{golden_dir}/ops_{dn}.npz")
    tol = 5e-5 if dn == "f32" else 1e-11
    for tag in ("a", "b"):
        g = lambda k: z[f"agcn_{tag}:{k}"]
        K = int(g("meta")[4])
        for ref_order in (False, True):
            y, c = O.agcn_fwd(g("x"), [g("s1"), g("s2")], g("w"), g("b"), K, reference_order=ref_order)
            assert relerr(y, g("y")) < tol
        y, c = O.agcn_fwd(g("x"), [g("s1"), g("s2")], g("w"), g("b"), K)
        dx, dS, dW, db = O.agcn_bwd(g("dy"), c)
        for a, b in ((dx, g("dx")), (dS[0], g("ds1")), (dS[1], g("ds2")), (dW, g("dw")), (db, g("db"))):
            assert relerr(a, b) < tol
    for tag in ("a", "b"):
        g = lambda k: z[f"cell_{tag}:{k}"]
        K = int(g("meta")[4])
        p = dict(gate_w=g("gw"), gate_b=g("gb"), update_w=g("uw"), update_b=g("ub"))
        hn, c = O.cell_fwd(g("x"), g("h"), [g("s1"), g("s2")], p, K)
        assert relerr(hn, g("hn")) < tol
        dx, dh, dS, gr = O.cell_bwd(g("dhn"), c)
        for a, b in ((dx, g("dx")), (dh, g("dh")), (dS[0], g("ds1")), (dS[1], g("ds2")),
                     (gr["gate_w"], g("dgw")), (gr["gate_b"], g("dgb")),
                     (gr["update_w"], g("duw")), (gr["update_b"], g("dub"))):
            assert relerr(a, b) < tol


def test_oracle_eval_metrics_match_reference_evaluate(golden_dir):
    """The oracle's restatement of evaluate()'s figures (model/traintest_MegaCRN.py:63-93, model/utils.py:126-160)
    against values computed by the reference's own statements (tests/golden/make_golden_utils.py)."""
    z = np.load(f"{golden_dir}/utils_f32.npz")
    B, T, N, D, nb = [int(v) for v in z["eval:meta"]]
    rows = []
    ypad = np.concatenate([z["eval:y"], np.repeat(z["eval:y"][-1:], nb * B - len(z["eval:y"]), axis=0)])   # loader padding
    for i in range(nb):
        y0 = ypad[i * B:(i + 1) * B, ..., :1].astype(np.float32)
        outs = (z[f"eval:output{i}"], None, z[f"eval:query{i}"], z[f"eval:pos{i}"], z[f"eval:neg{i}"])
        rows.append(O.eval_batch(outs, y0, SC_MEAN, SC_STD))
    np.testing.assert_allclose(np.array(rows), z["eval:per_batch"], rtol=2e-5)
    ep = O.eval_epoch(rows)
    assert abs(ep[0] - float(z["eval:mean_loss"])) < 1e-5 * abs(ep[0])
    logged = z["eval:logged"].reshape(-1)              # what the reference logs, 4 decimals
    np.testing.assert_allclose(np.array(ep[1:]), logged, atol=6e-5)


# ---- the PyTorch-CPU restatement timed by bench.py's cpu_baseline leg (oracle/megacrn_torch_cpu.py) -------------------
@pytest.mark.parametrize("name,dn", [("tiny", "f32"), ("odd", "f64"), ("layers2", "f32"), ("cheb2", "f64"), ("metrla", "f32")])
def test_torch_cpu_restatement_matches_reference(name, dn):
    """Same reference-generated vectors: eval forward, train-mode forward, loss, every gradient, grad norm, and the
    3-step Adam trajectory (torch.optim.Adam(lr=0.01, eps=1e-3) + clip_grad_norm_(5), traintest_MegaCRN.py:104,129)."""
    import torch
    from oracle import megacrn_torch_cpu as TC
    rec, Pn, m = load_case(name, dn)
    dt = torch.float32 if dn == "f32" else torch.float64
    t = lambda a: torch.tensor(np.asarray(a), dtype=dt)
    torch.set_num_threads(4)
    P = TC.make_params(Pn, dt)
    x, ycov, y = t(rec["x"]), t(rec["ycov"]), t(rec["labels"])
    with torch.no_grad():
        outs = TC.forward(P, x, ycov, cheb_k=m["cheb_k"], num_layers=m["num_layers"])
    for nm, o in zip(("output", "h_att", "query", "pos", "neg"), outs):
        assert relerr(o.numpy(), rec["eval:" + nm]) < TOL[dn], nm
    teacher = [bool(v) for v in rec["teacher"]]
    outs = TC.forward(P, x, ycov, y, teacher, m["cheb_k"], m["num_layers"])
    loss = TC.loss_terms(outs, y, SC_MEAN, SC_STD)
    loss.backward()
    assert abs(loss.item() - rec["train:loss"][0]) < 10 * TOL[dn] * abs(rec["train:loss"][0])
    for k, p in P.items():
        assert relerr(p.grad.numpy(), rec["g:" + k]) < GTOL[dn], k
    if "traj:loss" in rec:
        P = TC.make_params(Pn, dt)
        opt = torch.optim.Adam(list(P.values()), lr=0.01, eps=1e-3)
        got = [TC.train_step(P, opt, x, ycov, y, [bool(v) for v in rec["traj:teacher"][s]], SC_MEAN, SC_STD,
                             m["cheb_k"], m["num_layers"]) for s in range(3)]
        np.testing.assert_allclose(got, rec["traj:loss"], rtol=1e-4 if dn == "f32" else 1e-9)
