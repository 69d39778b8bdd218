"""LinearAppr auxiliaries (src/linpro.jl:181-204) and the index-based Heun guide of src/guip.jl:181-189 / src/ode.jl:98-113.

The reference's constructor cannot run as committed (kerneli reads an undefined `i`, and `b` is handed where only `_b`
exists for a LinearAppr); product and oracle restate it with the loop index it evidently means.  Tests:
  CPU  host C++ guide == oracle, bit for bit (host-only context); linearappr(Y, P) == oracle; for a LINEAR target the
       linearisation along any path is the target itself, so the Heun guide must agree with the Ralston-3 guide of the
       same LinPro auxiliary to O(dt^2), and with halved steps the difference must fall by ~4.
  GPU  guided paths / log-likelihoods of a Lorenz GuidedBridge with a LinearAppr auxiliary along a reference trajectory
       == the oracle; and the smoothing loop (SegChains) over such segments == bo_smooth_mcmc.
"""
import math

import numpy as np
import pytest

import bridgehip as bh
import oracle as o

LOR = dict(theta=(10.0, 20.0, 8 / 3), sigma=(3.0, 3.0, 3.0))       # test/smoothing.jl:19-20
LOR_PAR = [10.0, 20.0, 8 / 3, 3.0, 3.0, 3.0]


def lorenz_reference_path(tt, x0=(1.5, -1.5, 25.0)):
    """a smooth reference trajectory: the drift-only Euler solution (test/smoothing.jl:27 uses R3 of the drift)"""
    Y = np.zeros((len(tt), 3))
    y = np.array(x0)
    th = LOR_PAR
    for i in range(len(tt)):
        Y[i] = y
        if i + 1 < len(tt):
            y = y + np.array([th[0] * (y[1] - y[0]), y[0] * (th[1] - y[2]) - y[1], y[0] * y[1] - th[2] * y[2]]) * (tt[i + 1] - tt[i])
    return Y


def test_linearappr_and_heun_guide_host_equals_oracle():
    hctx = bh.Context(-1)                                   # host-only: coefficients can be computed and read back
    tt = np.linspace(0.0, 0.4, 81)
    Y = lorenz_reference_path(tt)
    P = bh.Lorenz(LOR["theta"], LOR["sigma"])
    v, hT = np.array([2.0, -1.0, 24.0]), 0.3 * np.eye(3) + 0.05 * np.ones((3, 3))
    Po = bh.GuidedBridge(tt, P, bh.linearappr(Y), v, hT, ctx=hctx)
    B, b, S = o.linearappr(o.MODEL_LORENZ, 3, 3, LOR_PAR, tt, Y)
    assert np.array_equal(Po.Pt.B, B) and np.array_equal(Po.Pt.b, b) and np.array_equal(Po.Pt.Sigma, S)
    assert np.array_equal(B[5], [[-10.0, 10.0, 0.0], [20.0 - Y[5, 2], -1.0, -Y[5, 0]], [Y[5, 1], Y[5, 0], -8 / 3]])   # src/Models.jl:49-53
    Hd, V = o.gp_hv_heuni(tt, 3, 3, Y, B, b, S, v, hT)
    assert np.array_equal(Po.Hd, Hd) and np.array_equal(Po.V, V)
    assert np.array_equal(Hd[-1], hT) and np.array_equal(V[-1], v)
    with pytest.raises(bh.BridgeError, match="traceB"):
        bh.lptilde(Po, [0.0, 0.0, 0.0])
    # pendulum: bderiv with cos (src/Models.jl:81-84)
    Pp = bh.GuidedBridge(np.linspace(0, 1, 21), bh.Pendulum(4.0, 0.5), bh.linearappr(np.stack([np.linspace(0, 1, 21), np.ones(21)], 1)),
                         [0.5, 0.2], 0.1 * np.eye(2), ctx=hctx)
    assert Pp.Pt.B[3][1, 0] == -4.0 * math.cos(3 / 20) and Pp.Pt.B[3][0, 1] == 1.0
    # processes without bderiv in the reference are refused
    with pytest.raises(bh.BridgeError, match="bderiv"):
        bh.GuidedBridge(np.linspace(0, 1, 11), bh.FitzhughDiffusion(0.1, 0.0, 1.5, 0.8, 0.3), bh.linearappr(np.zeros((11, 2))), [0.0, 0.0],
                        np.eye(2), ctx=hctx)


def test_heun_guide_of_a_linear_target_converges_to_the_r3_guide():
    hctx = bh.Context(-1)
    Bm = np.array([[-1.0, 0.3], [-0.2, -0.8]])
    sg = np.array([[0.8, 0.1], [-0.3, 0.6]])
    P = bh.LinPro(Bm, [0.1, -0.2], sg)
    v, hT = np.array([0.4, 0.1]), 0.2 * np.eye(2)
    errs = []
    for n in (101, 201, 401):
        tt = np.linspace(0, 1.0, n)
        Y = np.stack([np.sin(3 * tt), np.cos(2 * tt)], 1)        # ANY path: a linear drift is its own linearisation
        la = bh.GuidedBridge(tt, P, bh.linearappr(Y), v, hT, ctx=hctx)
        r3 = bh.GuidedBridge(tt, P, P, v, hT, ctx=hctx)
        errs.append(max(np.abs(la.Hd - r3.Hd).max(), np.abs(la.V - r3.V).max()))
    assert errs[0] < 1e-3 and errs[1] < errs[0] / 3.8 and errs[2] < errs[1] / 3.8      # second order: 5.8e-4, 1.5e-4, 3.7e-5


@pytest.mark.gpu
def test_lorenz_guidedbridge_with_linearappr_on_device():
    ctx = bh.default_context(0)
    tt = np.linspace(0.0, 0.3, 121)
    Y = lorenz_reference_path(tt)
    P = bh.Lorenz(LOR["theta"], LOR["sigma"])
    v, hT = Y[-1] + np.array([0.5, -0.3, 0.2]), 0.5 * np.eye(3)
    Po = bh.GuidedBridge(tt, P, bh.linearappr(Y), v, hT, ctx=ctx)
    apar = o.linearappr_par(tt, Y, Po.Pt.B, Po.Pt.b, Po.Pt.Sigma)
    ref = o.proposal_hv(tt, 3, 3, o.MODEL_LORENZ, LOR_PAR, o.AUX_LINEARAPPR, apar, Po.Hd, Po.V)
    X, W, ll = bh.sample_solve(Y[0], Po, 130, seed=8, store_W=True)
    Xh, Wh, llh = X.paths(), W.paths(), ll.cpu().numpy()
    for p in (0, 64, 129):
        Xo = o.solve_guided(ref, Y[0], Wh[p])
        assert np.array_equal(Xh[p], Xo)
        assert llh[p] == o.llikelihood(ref, Xo)
    # the guide pulls: proposals end near v (free end with hT = 0.5 I), far closer than the unguided process would
    assert np.abs(Xh[:, -1, :].mean(0) - v).max() < 1.0


@pytest.mark.gpu
def test_smoothing_loop_with_linearappr_segments():
    """test/smoothing.jl:73-92: segments whose auxiliaries are LinearAppr's along a reference solution, linked by gpupdate;
    the joint MH of supplements/smoothing/smoothing.jl:165-213 over them == the oracle twin"""
    ctx = bh.default_context(0)
    m, M = 3, 40
    tgrid = np.linspace(0.0, 0.24, m * M + 1)
    Yall = lorenz_reference_path(tgrid)
    P = bh.Lorenz(LOR["theta"], LOR["sigma"])
    L, Sig = np.eye(3), np.eye(3)
    rng = np.random.default_rng(1)
    obs = Yall[::M] + rng.standard_normal((m + 1, 3))
    H, v = bh.gpupdate(np.diag([np.inf] * 3), np.zeros(3), L, Sig, obs[m])
    segs, refs = [None] * m, [None] * m
    for i in range(m - 1, -1, -1):
        tt = tgrid[i * M:(i + 1) * M + 1].copy()
        Y = Yall[i * M:(i + 1) * M + 1]
        segs[i] = bh.GuidedBridge(tt, P, bh.linearappr(Y), v, H, ctx=ctx)
        apar = o.linearappr_par(tt, Y, segs[i].Pt.B, segs[i].Pt.b, segs[i].Pt.Sigma)
        refs[i] = o.proposal_hv(tt, 3, 3, o.MODEL_LORENZ, LOR_PAR, o.AUX_LINEARAPPR, apar, segs[i].Hd, segs[i].V)
        H, v = bh.gpupdate(segs[i], L, Sig, obs[i])
    mu, chol = v, np.linalg.cholesky((H + H.T) / 2)
    iters = 6
    w_new = np.sqrt(np.full(iters, 0.05))          # test/smoothing.jl:5  rho = 0.05
    w_old = np.sqrt(1 - w_new ** 2)
    sc = bh.SegChains(segs, mu, chol, 96, seed=2, mcnext=True)
    sc.step(w_old, w_new)
    ll, acc, y0 = sc.state()
    for p in (0, 95):
        r = o.smooth_mcmc(refs, mu, chol, w_old, w_new, 2, p, stats=True)
        for i in range(m):
            X, W = sc.paths(i, p, 1)
            assert np.array_equal(X[0], r["X"][i]) and np.array_equal(W[0], r["W"][i])
            mean, m2, cnt = sc.mcstats(i, p)
            assert np.array_equal(mean, r["mean"][i]) and np.array_equal(m2, r["m2"][i])
        assert np.array_equal(ll[:, p], r["ll"]) and acc[p] == r["acc"] and np.array_equal(y0[p], r["y0"])
    assert acc.sum() > 0


@pytest.mark.gpu
def test_adaptive_relinearisation_of_the_smoothing_segments():
    """supplements/smoothing/smoothing.jl:130-160: every `adaptit` iterations the LinearAppr auxiliaries are re-linearised
    around the running means, the chain of GuidedBridge's is rebuilt with gpupdate and the sampler continues.
    (i) handing over re-built but IDENTICAL proposals changes nothing, bit for bit; (ii) after a real adaptation the stored
    log-likelihoods are those of the current paths under the NEW proposals (== oracle), the chains continue and accept."""
    ctx = bh.default_context(0)
    m, M, n = 3, 40, 256
    tgrid = np.linspace(0.0, 0.24, m * M + 1)
    Yall = lorenz_reference_path(tgrid, x0=(2.5, -0.5, 23.0))          # a deliberately poor first linearisation point
    P = bh.Lorenz(LOR["theta"], LOR["sigma"])
    L, Sig = np.eye(3), 0.25 * np.eye(3)
    rng = np.random.default_rng(4)
    truth = lorenz_reference_path(tgrid)
    obs = truth[::M] + 0.5 * rng.standard_normal((m + 1, 3))
    HT, vT = bh.gpupdate(np.diag([np.inf] * 3), np.zeros(3), L, Sig, obs[m])

    def build(paths):
        H, v, segs = HT, vT, [None] * m
        for i in range(m - 1, -1, -1):
            segs[i] = bh.GuidedBridge(tgrid[i * M:(i + 1) * M + 1].copy(), P, bh.linearappr(paths[i]), v, H, ctx=ctx)
            H, v = bh.gpupdate(segs[i], L, Sig, obs[i])
        return segs, v, H
    first = [Yall[i * M:(i + 1) * M + 1] for i in range(m)]
    segs, mu, H0 = build(first)
    chol = np.linalg.cholesky((H0 + H0.T) / 2)
    a = bh.SegChains(segs, mu, chol, n, seed=9, pooled=True)
    b = bh.SegChains(build(first)[0], mu, chol, n, seed=9, pooled=True)
    a.step(0.95, math.sqrt(1 - 0.95 ** 2), 4)
    b.step(0.95, math.sqrt(1 - 0.95 ** 2), 2)
    b.set_proposals(build(first)[0])                                   # (i)
    b.step(0.95, math.sqrt(1 - 0.95 ** 2), 2)
    for u, w in zip(a.state(), b.state()):
        assert np.array_equal(u, w)
    assert np.array_equal(a.paths(1)[0], b.paths(1)[0])
    # (ii)
    means = [a.pooled_stats(i)[0] for i in range(m)]
    mu2, H2 = a.adapt(P, L, Sig, obs, HT, vT)
    ll, acc0, _ = a.state()
    for i in range(m):
        assert np.array_equal(a.pos[i].Pt.xx, means[i])
        apar = o.linearappr_par(a.pos[i].tt, means[i], a.pos[i].Pt.B, a.pos[i].Pt.b, a.pos[i].Pt.Sigma)
        ref = o.proposal_hv(a.pos[i].tt, 3, 3, o.MODEL_LORENZ, LOR_PAR, o.AUX_LINEARAPPR, apar, a.pos[i].Hd, a.pos[i].V)
        X, _ = a.paths(i, 17, 1)
        assert ll[i, 17] == o.llikelihood(ref, X[0])
    a.step(0.95, math.sqrt(1 - 0.95 ** 2), 6)
    _, acc1, _ = a.state()
    assert (acc1 - acc0).sum() > 0 and np.isfinite(a.state()[0]).all()


def test_linearnoiseappr_host_equals_oracle():
    """LinearNoiseAppr (src/guip.jl:114-146): deterministic path by R3 (forward / backward / nothing), slope coefficients, and the
    GuidedBridge built on it (index-based Heun) -- host C++ == oracle bit for bit; properties: B = 0 makes Hd(t) = hT + a (T - t)
    and V the trapezoidal integral of the slopes."""
    hctx = bh.Context(-1)
    tt = np.linspace(0.0, 0.3, 61) ** 1.0
    P = bh.Lorenz(LOR["theta"], LOR["sigma"])
    v, hT = np.array([2.0, -1.0, 24.0]), 0.3 * np.eye(3)
    for direction, dnum in (("backward", -1), ("forward", 1), ("nothing", 0)):
        Po = bh.GuidedBridge(tt, P, bh.LinearNoiseAppr(tt, P, v, None, direction), v, hT, ctx=hctx)
        Y = o.lna_path(o.MODEL_LORENZ, 3, LOR_PAR, tt, v, dnum)
        assert np.array_equal(Po.Pt.Y, Y)
        xx, B, b, S = o.lna_coeffs(o.MODEL_LORENZ, 3, 3, LOR_PAR, tt, Y)
        Hd, V = o.gp_hv_heuni(tt, 3, 3, xx, B, b, S, v, hT)
        assert np.array_equal(Po.Hd, Hd) and np.array_equal(Po.V, V)
        if dnum == -1:
            assert np.array_equal(Y[-1], v)
            assert np.array_equal(b[0], b[1]) and np.array_equal(b[5], (Y[5] - Y[4]) / (tt[5] - tt[4]))    # max(i, 2)
        if dnum == 1:
            assert np.array_equal(Y[0], v)
        a = np.diag(np.array(LOR["sigma"]) ** 2)
        assert np.allclose(Hd[0], hT + a * (tt[-1] - tt[0]), rtol=1e-13)
        assert np.allclose(V[0], v - sum(0.5 * (b[i] + b[i + 1]) * (tt[i + 1] - tt[i]) for i in range(len(tt) - 1)), rtol=1e-12, atol=1e-12)
    with pytest.raises(bh.BridgeError, match="Lorenz, Pendulum, LinPro, Wiener"):
        bh.GuidedBridge(np.linspace(0, 1, 11), bh.FitzhughDiffusion(0.1, 0.0, 1.5, 0.8, 0.3), bh.LinearNoiseAppr(x=[0.0, 0.0], direction="backward"),
                        [0.0, 0.0], np.eye(2), ctx=hctx)
