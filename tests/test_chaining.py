"""The callers either side of the hot path: segment chaining with gpupdate
(src/guip.jl:221-243, test/smoothing.jl:73-92), the inverse map innovations! (src/euler.jl:358-376)
and the scripts' output files (partialbridge_fitzhugh.jl:180-208)."""
import os

import numpy as np
import pytest
import torch

import bridgehip as bh
import oracle as o
import problems


# --------------------------------------------------------------------------- CPU
def test_gpupdate_matches_oracle_and_kalman_form():
    rng = np.random.default_rng(2)
    for d, m in ((1, 1), (2, 1), (3, 2), (3, 3)):
        A = rng.standard_normal((d, d))
        H = A @ A.T + np.eye(d)
        V = rng.standard_normal(d)
        L = rng.standard_normal((m, d))
        S = 0.3 * np.eye(m)
        v = rng.standard_normal(m)
        Hn, Vn = bh.gpupdate(H, V, L, S, v)
        Ho, Vo = o.gpupdate(H, V, L, S, v)
        assert np.array_equal(Hn, Ho) and np.array_equal(Vn, Vo)               # product host == oracle, bit for bit
        K = H @ L.T @ np.linalg.inv(S + L @ H @ L.T)                             # the Kalman / conditioning form
        assert np.allclose(Hn, H - K @ L @ H, atol=1e-12)
        assert np.allclose(Vn, V + K @ (v - L @ V), atol=1e-12)
    # Hdiamond = Inf*I (no information yet): the observation alone  src/guip.jl:222-225
    Hn, Vn = bh.gpupdate(np.diag([np.inf, np.inf]), [0.0, 0.0], np.eye(2), 0.5 * np.eye(2), [1.0, -2.0])
    assert np.allclose(Hn, 0.5 * np.eye(2)) and np.allclose(Vn, [1.0, -2.0])


def test_output_files(tmp_path):
    tt = np.linspace(0, 1, 101)
    XX = [np.stack([np.sin(tt + s), np.cos(tt + s)], axis=1) for s in (0, 1)]
    fn = tmp_path / "iterates.csv"
    bh.write_iterates_csv(fn, XX, [0, 1000], tt, every=50)
    lines = open(fn).read().splitlines()
    assert lines[0] == "iteration, time, component, value "                     # partialbridge_fitzhugh.jl:184
    assert len(lines) == 1 + 2 * 3 * 2                                           # 2 iterates x 3 time points x 2 components
    it, t, comp, val = lines[4].split(",")
    assert (int(it), float(t), int(comp)) == (0, tt[50], 2) and float(val) == XX[0][50, 1]
    pct = bh.write_info(tmp_path / "info.txt", "linearised_end", "extreme", 1000, 100, [-0.5, -0.6], 2.0, 1.1, [[1e-10]], [[1.0, 0.0]],
                        1 / 5000, 0.9, 437)
    txt = open(tmp_path / "info.txt").read()
    assert pct == 44.0 and "Average acceptance percentage: 44.0" in txt and "Choice of auxiliary process: linearised_end" in txt


# --------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("name", ["linpro2_guidedbridge", "fhn2_nuh_full", "ouproc_nuh", "linpro3_guidedbridge"])
def test_innovations_inverts_solve(name):
    ctx = bh.default_context(0)
    c = [k for k in problems.cases(201) if k.name == name][0]
    Po, ref = c.bh_proposal(bh, ctx), c.oracle_proposal()
    P = 70
    X, W, _ = bh.sample_solve(c.x0, Po, P, seed=9, store_W=True)
    W2 = bh.innovations(bh.Euler(), X, Po)
    Xh, W2h = X.paths(), W2.paths()
    for p in (0, 64, 69):
        assert np.array_equal(W2h[p], o.innovations(ref, Xh[p]))                 # bit-exact vs the oracle
    # round trip: the recovered W drives the same path (up to the rounding of the inverse map)
    X3 = bh.solve(bh.Euler(), c.x0, W2, Po)
    assert float((X3.data[:-1] - X.data[:-1]).abs().max()) < 1e-9 * (1 + float(X.data.abs().max()))
    assert float((W2.data[:-1] - W.data[:-1]).abs().max()) < 1e-8 * (1 + float(W.data.abs().max()))


@pytest.mark.gpu
def test_innovations_plain_process_and_errors():
    ctx = bh.default_context(0)
    tt = np.linspace(0, 1.0, 201)
    lor = bh.Lorenz((10.0, 28.0, 8 / 3), (3.0, 3.0, 3.0))
    proc = bh.PlainProcess(tt, lor, ctx=ctx)
    W = bh.sample(tt, bh.Wiener(3), npaths=33, seed=4, ctx=ctx)
    X = bh.solve(bh.Euler(), [1.0, 0.0, 0.0], W, proc)
    W2 = bh.innovations(bh.Euler(), X, proc)
    Xh = X.paths()
    ref = o.innovations(None, Xh[5], model=o.MODEL_LORENZ, d=3, mp=3, par=[10.0, 28.0, 8 / 3, 3.0, 3.0, 3.0], tt=tt)
    assert np.array_equal(W2.paths()[5], ref)
    assert float((W2.data - W.data).abs().max()) < 1e-10
    c = [k for k in problems.cases(51) if k.name == "fhn_partialbridge_first"][0]      # sigma = (0, sigma): not invertible
    Po = c.bh_proposal(bh, ctx)
    Xf, _, _ = bh.sample_solve(c.x0, Po, 8, seed=1)
    with pytest.raises(bh.BridgeError, match="square"):
        bh.innovations(bh.Euler(), Xf, Po)


@pytest.mark.gpu
def test_chained_guided_bridge_segments_smoothing():
    """test/smoothing.jl:73-92: m segments, each a GuidedBridge whose (Hdiamond, v) at its right end comes from
    gpupdate of the next segment; forward sampling chains the endpoints (solve! returns yy[N])."""
    ctx = bh.default_context(0)
    rng = np.random.default_rng(3)
    d, m, M, npaths = 2, 4, 50, 512
    B = np.array([[-1, 0.1], [-0.2, -1]])
    sig = 2 * np.array([[-0.212887, 0.0687025], [0.193157, 0.388997]])
    P = bh.LinPro(B, [0.0, 0.0], sig)
    Pt = bh.LinPro(0.8 * B, [0.0, 0.0], sig)
    L, Sig = np.array([[1.0, 0.0]]), np.array([[0.05]])
    tgrid = np.linspace(0, 2.0, m * M + 1)
    obs = rng.standard_normal((m + 1, 1))                      # one partial observation at every segment boundary
    segs = [None] * m
    H = np.diag([np.inf, np.inf])
    H, v = bh.gpupdate(H, np.zeros(d), np.eye(d), 0.5 * np.eye(d), [obs[m, 0], 0.0])     # a proper prior at the right end
    Ho, vo = H.copy(), v.copy()
    for i in range(m - 1, -1, -1):
        tt = tgrid[i * M:(i + 1) * M + 1].copy()
        segs[i] = bh.GuidedBridge(tt, P, Pt, v, H, ctx=ctx)
        H, v = bh.gpupdate(segs[i], L, Sig, obs[i])
        # the same backward recursion through the oracle gives identical coefficients
        apar = o.linpro_par(0.8 * B, [0.0, 0.0], sig)
        Hd_o, V_o = o.gp_hv(tt, d, d, o.AUX_LINPRO, apar, vo, Ho)
        assert np.array_equal(segs[i].Hd, Hd_o) and np.array_equal(segs[i].V, V_o)
        Ho, vo = o.gpupdate(Hd_o[0], V_o[0], L, Sig, obs[i])
        assert np.array_equal(H, Ho) and np.array_equal(v, vo)
    # forward: y = x0; for i in 1:m: sample!(WW[i]); y = solve!(Euler(), XX[i], y, WW[i], Po[i])
    x0 = np.array([0.3, -0.2])
    y, XX, WW, lls = x0, [], [], []
    for i in range(m):
        W = bh.sample(segs[i].tt, bh.Wiener(d), npaths=npaths, seed=21, iter=i, ctx=ctx)
        Y = bh.EnsemblePath(segs[i].tt, d, npaths, ctx)
        ll = ctx.empty(npaths)
        y = bh.solve_(bh.Euler(), Y, y, W, segs[i], ll=ll).contiguous()      # endpoints [d, npaths] feed the next segment
        XX.append(Y)
        WW.append(W)
        lls.append(ll)
    # continuity at the joints and parity with the oracle for one path through all segments
    for i in range(m - 1):
        assert torch.equal(XX[i].data[-1], XX[i + 1].data[0])
    p = 77
    yo = x0
    par = o.linpro_par(B, [0.0, 0.0], sig)
    for i in range(m):
        ref = o.proposal_hv(segs[i].tt, d, d, o.MODEL_LINPRO, par, o.AUX_LINPRO, apar, segs[i].Hd, segs[i].V)
        Wp = WW[i].paths(p, 1)[0]
        Xo = o.solve_guided(ref, yo, Wp)
        assert np.array_equal(XX[i].paths(p, 1)[0], Xo)
        assert float(lls[i][p]) == o.llikelihood(ref, Xo)
        yo = Xo[-1]
    # mcnext!-style pointwise statistics over the ensemble of the last segment (supplements/smoothing.jl:211-213)
    mc = bh.mcstart(np.zeros((M + 1, d)))
    Xlast = XX[-1].paths()
    for q in range(64):
        mc = bh.mcnext(mc, Xlast[q])
    mean, cov = bh.mcstats(mc)
    assert np.allclose(mean, Xlast[:64].mean(0)) and cov.shape == (M + 1, d, d)
