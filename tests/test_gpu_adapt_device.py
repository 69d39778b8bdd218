"""Guide pre-computation on the device, one guide per chain: the adaptation block of
supplements/smoothing/smoothing.jl:130-160 run for every chain of the ensemble at once (bhip_segchains_adapt_device,
bhip_guide_kernel.h) against the oracle's single-chain restatement of the whole loop (bo_smooth_adaptive).

Lorenz (polynomial drift and Jacobian): every chain's guide, pi0, paths, Wiener paths, log-likelihoods, acceptance counts
and mcnext! states are compared bit for bit.  Pendulum (sin / cos in drift and Jacobian): bit for bit as well since oracle,
host and kernels share one fdlibm-form sin / cos (bhip_trig.h, bo_sin / bo_cos).
"""
import math

import numpy as np
import pytest

import bridgehip as bh
import oracle as o

pytestmark = pytest.mark.gpu

LOR = dict(theta=(10.0, 20.0, 8 / 3), sigma=(3.0, 3.0, 3.0))
LOR_PAR = [10.0, 20.0, 8 / 3, 3.0, 3.0, 3.0]


def lorenz_drift_path(tt, x0):
    Y = np.zeros((len(tt), 3))
    y = np.array(x0, dtype=np.float64)
    th = LOR_PAR
    for i in range(len(tt)):
        Y[i] = y
        if i + 1 < len(tt):
            y = y + np.array([th[0] * (y[1] - y[0]), y[0] * (th[1] - y[2]) - y[1], y[0] * y[1] - th[2] * y[2]]) * (tt[i + 1] - tt[i])
    return Y


def lorenz_setup(ctx, m, M, L, Sig, seed):
    tgrid = np.linspace(0.0, 0.08 * m, m * M + 1)
    truth = lorenz_drift_path(tgrid, (1.5, -1.5, 25.0))
    first = lorenz_drift_path(tgrid, (2.5, -0.5, 23.0))          # a deliberately poor first linearisation
    rng = np.random.default_rng(seed)
    L = np.atleast_2d(np.asarray(L, dtype=np.float64))
    obs = truth[::M] @ L.T + 0.5 * rng.standard_normal((m + 1, L.shape[0]))
    P = bh.Lorenz(LOR["theta"], LOR["sigma"])
    HT, vT = bh.gpupdate(1e3 * np.eye(3), np.zeros(3), L, Sig, obs[m])          # smoothing.jl:75: piH*I prior, last observation
    tts = np.stack([tgrid[i * M:(i + 1) * M + 1] for i in range(m)])
    Y0 = np.stack([first[i * M:(i + 1) * M + 1] for i in range(m)])
    H, v, segs = HT, vT, [None] * m
    for i in range(m - 1, -1, -1):
        segs[i] = bh.GuidedBridge(tts[i].copy(), P, bh.linearappr(Y0[i]), v, H, ctx=ctx)
        H, v = bh.gpupdate(segs[i], L, Sig, obs[i])
    return dict(P=P, tts=tts, Y0=Y0, obs=obs, HT=HT, vT=vT, segs=segs, mu=v, H0=H, L=L, Sig=np.atleast_2d(Sig))


@pytest.mark.parametrize("L,Sig,hwindow", [(np.eye(3), 0.25 * np.eye(3), 0), ([[1.0, 0.0, 0.0]], [[0.25]], 0), (np.eye(3), 0.25 * np.eye(3), 3)],
                         ids=["full-obs", "partial-obs", "smoothmean"])
def test_per_chain_adaptation_on_device_equals_the_reference_loop_chain_by_chain(L, Sig, hwindow):
    ctx = bh.default_context(0)
    m, M, n = 3, 40, 192
    S = lorenz_setup(ctx, m, M, L, Sig, seed=4)
    chol = o.chol_lower(S["H0"])
    adaptit, iters = 5, 12
    w_new = np.sqrt(np.full(iters, 0.1)); w_old = np.sqrt(1 - w_new ** 2)
    sc = bh.SegChains(S["segs"], S["mu"], chol, n, seed=21, mcnext=True)
    sc.step(w_old[:adaptit - 1], w_new[:adaptit - 1])
    # iteration `adaptit` starts with the adaptation (it % adaptit == 0), newblock = true, doaccept = (it == adaptit)
    sc.adapt_device(S["L"], S["Sig"], S["obs"][:m], S["HT"], S["vT"], hwindow=hwindow, newblock=True, doaccept=True)
    ll_ad = sc.state()[0].copy()
    guides = {p: [sc.chain_guide(i, p) for i in range(m)] for p in (0, 77, n - 1)}
    sc.step(w_old[adaptit - 1:2 * adaptit - 1], w_new[adaptit - 1:2 * adaptit - 1])
    sc.adapt_device(S["L"], S["Sig"], S["obs"][:m], S["HT"], S["vT"], hwindow=hwindow, newblock=True, doaccept=False)   # it = 2*adaptit
    sc.step(w_old[2 * adaptit - 1:], w_new[2 * adaptit - 1:])
    ll, acc, y0 = sc.state()
    for p in (0, 77, n - 1):
        # the state right after the first adaptation: the oracle stopped one iteration earlier has the same means; its guides
        # at iteration `adaptit` are what bo_smooth_adaptive builds there -- compare through a run that ends AT the adaptation
        r1 = o.smooth_adaptive(o.MODEL_LORENZ, 3, 3, LOR_PAR, S["tts"], S["Y0"], S["L"], S["Sig"], S["obs"][:m], S["HT"], S["vT"],
                               w_old[:adaptit], w_new[:adaptit], adaptit, 10 ** 6, 21, p, hwindow=hwindow)
        for i in range(m):
            g = guides[p][i]
            Hd, V = r1["Hd"][i], r1["V"][i]
            assert np.array_equal(g["G"][:, 10:13], V[:-1])
            assert np.array_equal(g["G"][:, 9], [_det3(h) for h in Hd[:-1]])
            assert np.array_equal(g["G"][:, :9], np.stack([_cof3(h) for h in Hd[:-1]]))
        assert np.array_equal(guides[p][0]["mu"], r1["mu"]) and np.array_equal(guides[p][0]["chol"], o.chol_lower(r1["H"]))
        # the whole run, two adaptations
        r = o.smooth_adaptive(o.MODEL_LORENZ, 3, 3, LOR_PAR, S["tts"], S["Y0"], S["L"], S["Sig"], S["obs"][:m], S["HT"], S["vT"],
                              w_old, w_new, adaptit, 10 ** 6, 21, p, hwindow=hwindow)
        for i in range(m):
            X, W = sc.paths(i, p, 1)
            assert np.array_equal(X[0], r["X"][i]) and np.array_equal(W[0], r["W"][i])
            mean, m2, cnt = sc.mcstats(i, p)
            assert np.array_equal(mean, r["mean"][i]) and np.array_equal(m2, r["m2"][i]) and cnt == iters
        assert np.array_equal(ll[:, p], r["ll"]) and acc[p] == r["acc"] and np.array_equal(y0[p], r["y0"])
    assert np.isfinite(ll_ad).all() and np.isfinite(ll).all()
    # iteration `adaptit` accepted everywhere (doaccept), so every chain has at least one acceptance
    assert (acc >= 1).all()
    # the chains' guides differ from each other (own means) -- and from the shared first guide
    assert not np.array_equal(guides[0][1]["G"], guides[77][1]["G"])


def _cof3(A):
    a = lambda i, j: A[i - 1, j - 1]
    return np.array([a(2, 2) * a(3, 3) - a(2, 3) * a(3, 2), a(1, 3) * a(3, 2) - a(1, 2) * a(3, 3), a(1, 2) * a(2, 3) - a(1, 3) * a(2, 2),
                     a(2, 3) * a(3, 1) - a(2, 1) * a(3, 3), a(1, 1) * a(3, 3) - a(1, 3) * a(3, 1), a(1, 3) * a(2, 1) - a(1, 1) * a(2, 3),
                     a(2, 1) * a(3, 2) - a(2, 2) * a(3, 1), a(1, 2) * a(3, 1) - a(1, 1) * a(3, 2), a(1, 1) * a(2, 2) - a(1, 2) * a(2, 1)])


def _det3(A):
    a = A.T.ravel()      # column-major
    c0, c1, c2 = a[4] * a[8] - a[5] * a[7], a[5] * a[6] - a[3] * a[8], a[3] * a[7] - a[4] * a[6]
    return a[0] * c0 + a[1] * c1 + a[2] * c2


def test_adaptation_improves_the_proposals_and_host_route_agrees_for_identical_means():
    """(i) with every chain given the SAME linearisation path the device-built per-chain guides equal the host-built shared
    guide bit for bit (the ensemble then continues identically on either route); (ii) after adapting around their own means
    the chains accept more often than with the poor first linearisation."""
    ctx = bh.default_context(0)
    m, M, n = 3, 40, 512
    S = lorenz_setup(ctx, m, M, np.eye(3), 0.25 * np.eye(3), seed=4)
    chol = o.chol_lower(S["H0"])
    rho = 0.95
    wo, wn = rho, math.sqrt(1 - rho ** 2)
    a = bh.SegChains(S["segs"], S["mu"], chol, n, seed=3, mcnext=True)
    a.step(wo, wn, 1)
    # after ONE iteration a chain's mean is its current path; chains that rejected still hold... different paths per chain.
    # (i): compare chain p's device guide with a host-built guide around chain p's mean
    a.adapt_device(S["L"], S["Sig"], S["obs"][:m], S["HT"], S["vT"], newblock=False)
    for p in (5, 300):
        H, v = S["HT"], S["vT"]
        for i in range(m - 1, -1, -1):
            Y = a.mcstats(i, p)[0]
            Po = bh.GuidedBridge(S["tts"][i].copy(), S["P"], bh.linearappr(Y), v, H, ctx=ctx)
            g = a.chain_guide(i, p)
            assert np.array_equal(g["B"], Po.Pt.B[:-1]) and np.array_equal(g["G"][:, 10:13], Po.V[:-1])
            assert np.array_equal(g["beta"], np.stack([Po.Pt.b[j] - ((Po.Pt.B[j][:, 0] * Po.Pt.xx[j][0] + Po.Pt.B[j][:, 1] * Po.Pt.xx[j][1]) + Po.Pt.B[j][:, 2] * Po.Pt.xx[j][2]) for j in range(M)]))
            assert np.array_equal(g["G"][:, :9], np.stack([_cof3(h) for h in Po.Hd[:-1]]))
            H, v = bh.gpupdate(Po, S["L"], S["Sig"], S["obs"][i])
        assert np.array_equal(a.chain_guide(0, p)["mu"], v) and np.array_equal(a.chain_guide(0, p)["chol"], o.chol_lower(H))
    # (ii)
    rho = 0.5                                                   # bolder moves: the quality of the guide shows in the acceptance
    wo, wn = rho, math.sqrt(1 - rho ** 2)
    b = bh.SegChains(S["segs"], S["mu"], chol, n, seed=3, mcnext=True)
    b.step(wo, wn, 40)
    acc_before = b.state()[1].copy()
    b.adapt_device(S["L"], S["Sig"], S["obs"][:m], S["HT"], S["vT"], newblock=True, doaccept=True)
    b.step(wo, wn, 1)
    acc_mid = b.state()[1].copy()
    assert np.array_equal(acc_mid - acc_before, np.ones(n, dtype=np.int64))      # doaccept
    b.step(wo, wn, 40)
    acc_after = b.state()[1] - acc_mid
    assert (40 - acc_after.mean()) < 0.7 * (40 - acc_before.mean()), (acc_before.mean(), acc_after.mean())   # fewer rejections
    assert np.isfinite(b.state()[0]).all()
    # back to shared proposals
    b.set_proposals(S["segs"])
    b.step(wo, wn, 2)
    assert np.isfinite(b.state()[0]).all()


def test_pendulum_per_chain_guides_bit_exact_with_the_shared_sin_cos():
    ctx = bh.default_context(0)
    m, M, n = 2, 50, 130            # ragged: the line kernel replicates the last chain in the unused lanes
    tgrid = np.linspace(0.0, 1.0, m * M + 1)
    P = bh.Pendulum(4.0, 0.5)
    par = [4.0, 0.5]
    L, Sig = np.array([[1.0, 0.0]]), np.array([[0.04]])
    rng = np.random.default_rng(2)
    truth = np.stack([0.8 * np.cos(2 * tgrid), -1.6 * np.sin(2 * tgrid)], 1)
    obs = truth[::M] @ L.T + 0.2 * rng.standard_normal((m + 1, 1))
    HT, vT = bh.gpupdate(1e2 * np.eye(2), np.zeros(2), L, Sig, obs[m])
    tts = np.stack([tgrid[i * M:(i + 1) * M + 1] for i in range(m)])
    Y0 = np.stack([np.zeros((M + 1, 2)) + [0.5, 0.0] for _ in range(m)])
    H, v, segs = HT, vT, [None] * m
    for i in range(m - 1, -1, -1):
        segs[i] = bh.GuidedBridge(tts[i].copy(), P, bh.linearappr(Y0[i]), v, H, ctx=ctx)
        H, v = bh.gpupdate(segs[i], L, Sig, obs[i])
    chol = o.chol_lower(H)
    iters, adaptit = 9, 4
    w_new = np.sqrt(np.full(iters, 0.2)); w_old = np.sqrt(1 - w_new ** 2)
    sc = bh.SegChains(segs, v, chol, n, seed=5, mcnext=True)
    sc.step(w_old[:adaptit - 1], w_new[:adaptit - 1])
    sc.adapt_device(L, Sig, obs[:m], HT, vT, newblock=True, doaccept=True)
    for p in (0, 129):
        r1 = o.smooth_adaptive(o.MODEL_PENDULUM, 2, 1, par, tts, Y0, L, Sig, obs[:m], HT, vT, w_old[:adaptit], w_new[:adaptit], adaptit, 10 ** 6, 5, p)
        for i in range(m):
            g = sc.chain_guide(i, p)
            assert np.array_equal(g["G"][:, 5:7], r1["V"][i][:-1])
            assert np.array_equal(g["G"][:, :4], r1["Hd"][i][:-1].transpose(0, 2, 1).reshape(-1, 4))
        assert np.array_equal(sc.chain_guide(0, p)["mu"], r1["mu"]) and np.array_equal(sc.chain_guide(0, p)["chol"], o.chol_lower(r1["H"]))
    sc.step(w_old[adaptit - 1:], w_new[adaptit - 1:])
    ll, acc, y0 = sc.state()
    assert np.isfinite(ll).all() and (acc >= 1).all()
    for p in (0, 64, 129):     # the whole run, chain by chain
        r = o.smooth_adaptive(o.MODEL_PENDULUM, 2, 1, par, tts, Y0, L, Sig, obs[:m], HT, vT, w_old, w_new, adaptit, adaptit + 1, 5, p)   # adaptmax: one adaptation
        for i in range(m):
            X, W = sc.paths(i, p, 1)
            assert np.array_equal(X[0], r["X"][i]) and np.array_equal(W[0], r["W"][i])
        assert np.array_equal(ll[:, p], r["ll"]) and acc[p] == r["acc"] and np.array_equal(y0[p], r["y0"])


def test_adapt_device_argument_checks():
    ctx = bh.default_context(0)
    S = lorenz_setup(ctx, 2, 20, np.eye(3), 0.25 * np.eye(3), seed=1)
    chol = o.chol_lower(S["H0"])
    sc = bh.SegChains(S["segs"], S["mu"], chol, 64, seed=1)            # no mcnext
    sc.step(0.9, math.sqrt(1 - 0.81), 1)
    with pytest.raises(bh.BridgeError, match="MCNEXT"):
        sc.adapt_device(S["L"], S["Sig"], S["obs"][:2], S["HT"], S["vT"])
    sc2 = bh.SegChains(S["segs"], S["mu"], chol, 64, seed=1, mcnext=True)
    with pytest.raises(bh.BridgeError, match="no iteration"):
        sc2.adapt_device(S["L"], S["Sig"], S["obs"][:2], S["HT"], S["vT"])
    with pytest.raises(bh.BridgeError, match="per-chain"):
        sc2.chain_guide(0, 0)


def test_linearnoiseappr_segments_and_their_per_chain_adaptation():
    """supplements/smoothing/smoothing.jl with initnu = :backward (:28,85): every segment's auxiliary is
    LinearNoiseAppr(tt_i, P, v, a, :backward); the adaptation replaces its deterministic path by the chain's running mean
    (:136-139).  Device ensemble == the oracle's single-chain loop, bit for bit, through two adaptations."""
    ctx = bh.default_context(0)
    m, M, n = 3, 40, 100            # not a multiple of the wave size: exercises the tails of the guide and path kernels
    tgrid = np.linspace(0.0, 0.24, m * M + 1)
    truth = lorenz_drift_path(tgrid, (1.5, -1.5, 25.0))
    rng = np.random.default_rng(7)
    L, Sig = np.eye(3), 0.25 * np.eye(3)
    obs = truth[::M] + 0.5 * rng.standard_normal((m + 1, 3))
    P = bh.Lorenz(LOR["theta"], LOR["sigma"])
    HT, vT = bh.gpupdate(1e3 * np.eye(3), np.zeros(3), L, Sig, obs[m])
    tts = np.stack([tgrid[i * M:(i + 1) * M + 1] for i in range(m)])
    H, v, segs = HT, vT, [None] * m
    for i in range(m - 1, -1, -1):
        segs[i] = bh.GuidedBridge(tts[i].copy(), P, bh.LinearNoiseAppr(tts[i], P, v, None, "backward"), v, H, ctx=ctx)
        H, v = bh.gpupdate(segs[i], L, Sig, obs[i])
    adaptit, iters = 4, 10
    w_new = np.sqrt(np.full(iters, 0.1)); w_old = np.sqrt(1 - w_new ** 2)
    sc = bh.SegChains(segs, v, o.chol_lower(H), n, seed=13, mcnext=True)
    sc.step(w_old[:adaptit - 1], w_new[:adaptit - 1])
    sc.adapt_device(L, Sig, obs[:m], HT, vT, newblock=True, doaccept=True)
    sc.step(w_old[adaptit - 1:2 * adaptit - 1], w_new[adaptit - 1:2 * adaptit - 1])
    sc.adapt_device(L, Sig, obs[:m], HT, vT, newblock=True, doaccept=False)
    sc.step(w_old[2 * adaptit - 1:], w_new[2 * adaptit - 1:])
    ll, acc, y0 = sc.state()
    Y0 = np.zeros((m, M + 1, 3))
    for p in (0, 63, n - 1):
        r = o.smooth_adaptive(o.MODEL_LORENZ, 3, 3, LOR_PAR, tts, Y0, L, Sig, obs[:m], HT, vT, w_old, w_new, adaptit, 10 ** 6, 13, p, lna=2)
        for i in range(m):
            X, W = sc.paths(i, p, 1)
            assert np.array_equal(X[0], r["X"][i]) and np.array_equal(W[0], r["W"][i])
            g = sc.chain_guide(i, p)
            assert np.array_equal(g["G"][:, 10:13], r["V"][i][:-1]) and np.array_equal(g["B"], np.zeros((M, 3, 3)))
        assert np.array_equal(ll[:, p], r["ll"]) and acc[p] == r["acc"] and np.array_equal(y0[p], r["y0"])
        assert np.array_equal(sc.chain_guide(0, p)["mu"], r["mu"])
    assert np.isfinite(ll).all() and (acc >= 1).all()


def test_linear_target_per_chain_guides_are_the_targets_own_guide():
    """bderiv of a LinPro is its B (src/linpro.jl:82): the linearisation along ANY path is the target itself, so every chain's
    device-built guide must agree with every other chain's and with the host guide of the same LinPro used as its own auxiliary
    (Ralston-3) up to the second-order Heun error; the log-likelihood ratio then vanishes and every proposal is accepted."""
    ctx = bh.default_context(0)
    m, M, n = 2, 200, 4096
    Bm = np.array([[-1.0, 0.3], [-0.2, -0.8]])
    sg = np.array([[0.8, 0.1], [-0.3, 0.6]])
    mu_ = np.array([0.1, -0.2])
    P = bh.LinPro(Bm, mu_, sg)
    tgrid = np.linspace(0.0, 1.0, m * M + 1)
    L, Sig = np.eye(2), 0.1 * np.eye(2)
    obs = np.array([[0.3, -0.1], [0.5, 0.2], [0.2, 0.4]])
    HT, vT = bh.gpupdate(1e3 * np.eye(2), np.zeros(2), L, Sig, obs[m])
    H, v, segs, r3 = HT, vT, [None] * m, [None] * m
    for i in range(m - 1, -1, -1):
        tt = tgrid[i * M:(i + 1) * M + 1].copy()
        Y = np.stack([np.sin(3 * tt), np.cos(2 * tt)], 1)                      # an arbitrary first linearisation path
        segs[i] = bh.GuidedBridge(tt, P, bh.linearappr(Y), v, H, ctx=ctx)
        r3[i] = bh.GuidedBridge(tt, P, P, v, H, ctx=bh.Context(-1))
        H, v = bh.gpupdate(segs[i], L, Sig, obs[i])
    sc = bh.SegChains(segs, v, o.chol_lower(H), n, seed=17, mcnext=True)
    sc.step(0.8, 0.6, 3)
    sc.adapt_device(L, Sig, obs[:m], HT, vT, newblock=False)
    g = {p: sc.chain_guide(m - 1, p) for p in (0, 1000, n - 1)}
    for p in (1000, n - 1):
        assert np.allclose(g[p]["G"], g[0]["G"], rtol=1e-11, atol=1e-13) and np.array_equal(g[p]["B"], g[0]["B"])
    assert np.array_equal(g[0]["B"][7], Bm)
    assert np.abs(g[0]["G"][:, 5:7] - r3[m - 1].V[:-1]).max() < 1e-4          # Heun vs Ralston-3 on the same ODE
    assert np.abs(g[0]["G"][:, :4] - np.stack([h.T.ravel() for h in r3[m - 1].Hd[:-1]])).max() < 1e-4
    acc0 = sc.state()[1].copy()
    sc.step(0.8, 0.6, 10)
    ll, acc, _ = sc.state()
    assert np.array_equal(acc - acc0, np.full(n, 10)) and np.abs(ll).max() < 1e-9


def test_mean_only_statistics_drive_the_same_adaptation():
    """BHIP_SEGCHAINS_MCNEXT_MEAN keeps the running means only (the adaptation reads nothing else): the chains, the means and the
    device-built guides are bit-identical to the full mcnext! run; asking for the second moments is an error"""
    ctx = bh.default_context(0)
    m, M, n = 2, 40, 96
    S = lorenz_setup(ctx, m, M, np.eye(3), 0.25 * np.eye(3), seed=4)
    chol = o.chol_lower(S["H0"])
    a = bh.SegChains(S["segs"], S["mu"], chol, n, seed=2, mcnext=True)
    b = bh.SegChains(S["segs"], S["mu"], chol, n, seed=2, mcnext_mean_only=True)
    for sc in (a, b):
        sc.step(0.9, math.sqrt(1 - 0.81), 4)
        sc.adapt_device(S["L"], S["Sig"], S["obs"][:m], S["HT"], S["vT"], newblock=True, doaccept=True)
        sc.step(0.9, math.sqrt(1 - 0.81), 3)
    for u, v in zip(a.state(), b.state()):
        assert np.array_equal(u, v)
    ga, gb = a.chain_guide(1, 50), b.chain_guide(1, 50)
    assert np.array_equal(ga["G"], gb["G"]) and np.array_equal(ga["B"], gb["B"])
    ma = a.mcstats(0, 50)[0]
    mean_b = np.empty((M + 1, 3)); cnt = bh.api.C.c_int64()
    ctx.check(ctx.lib.bhip_segchains_mcstats(b.h, 0, 50, bh.api._dptr(mean_b), None, bh.api.C.byref(cnt)))
    assert np.array_equal(ma, mean_b) and cnt.value == 7
    mb, m2b, cb = b.mcstats(0, 50)                                   # the mirror hands back what is kept: the means, no second moments
    assert np.array_equal(mb, ma) and m2b is None and cb == 7
    m2 = np.empty((M + 1, 9))
    assert ctx.lib.bhip_segchains_mcstats(b.h, 0, 50, None, bh.api._dptr(m2), None) == -4 and b"second moments" in ctx.lib.bhip_last_error(ctx.h)
