"""The C oracle against a second, independently written numpy restatement of the Julia sources (tests/julia_restate.py)
and against a 50-digit mpmath evaluation of the same recurrences (VERDICT r1 "weak" 1 / "next" 6).

Parity of the guided path with Bridge.jl itself stays UNPINNED at bit level (the reference stores no guided paths and
Julia cannot run here); these tests make a shared transcription slip much harder: two restatements written separately
from the Julia text must agree to rounding, and the fp64 arithmetic must sit within the stated tolerance of exact
arithmetic even in the stiff regime near T (M_i ~ 1e10).
"""
import math

import numpy as np
import pytest

import julia_restate as jr
import oracle as o
import problems

N = 161


def case(name, n=N):
    return [c for c in problems.cases(n) if c.name == name][0]


def rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return np.abs(a - b).max() / (1 + np.abs(b).max())


def build(c):
    """the SAME problem through the numpy restatement (constructed from the case's raw numbers, not through the oracle)"""
    if c.model == o.MODEL_FHN:
        P = jr.FitzhughDiffusion(*c.par)
        Pt = jr.FitzhughAuxEnd(P, c.v[0])
    else:
        d = c.d
        up = lambda a: o.uncm(a, d, d)
        P = jr.LinPro(up(c.par[:d * d]), c.par[d * d:d * d + d], up(c.par[d * d + d:]))
        ap = np.asarray(c.apar, float)
        Pt = jr.LinPro(up(ap[:d * d]), ap[d * d:d * d + d], up(ap[d * d + d:]))
    if c.kind == o.GUIDE_LMMU:
        return jr.PartialBridge(c.tt, P, Pt, c.L, c.v, c.Sigma)
    if c.kind == o.GUIDE_HV:
        return jr.GuidedBridge(c.tt, P, Pt, c.v, c.hT)
    return jr.PartialBridgeNuH(c.tt, P, Pt, c.L, c.v, c.eps, c.Sigma)


NAMES = ["fhn_partialbridge_first", "fhn_partialbridge_extreme", "ou_guidedbridge", "ou_guidedbridge_free_end",
         "linpro2_guidedbridge", "linpro3_guidedbridge", "linpro3_partial_m2", "fhn_nuh"]


@pytest.mark.parametrize("name", NAMES)
def test_guide_coefficients_two_restatements_agree(name):
    c = case(name)
    if name == "fhn_nuh":      # its auxiliary is "linearised_end" at v = -1 as well (tests/problems.py)
        pass
    Po = build(c)
    g = c.oracle_guide()
    if c.kind == o.GUIDE_LMMU:
        # row N-1 holds inv(Sigma) = 1e10 (never read by the loops): compare relative to each matrix
        for i in range(len(c.tt)):
            assert rel(Po.L[i], g["L"][i]) < 1e-12 and rel(Po.mu[i], g["mu"][i]) < 1e-12
            assert np.abs(Po.M[i] - g["M"][i]).max() <= 1e-9 * np.abs(g["M"][i]).max()
    elif c.kind == o.GUIDE_HV:
        assert rel(np.array(Po.Hd), g["Hd"]) < 1e-12 and rel(np.array(Po.V), g["V"]) < 1e-12
    else:
        for i in range(len(c.tt)):
            assert np.abs(Po.H[i] - g["H"][i]).max() <= 1e-9 * np.abs(g["H"][i]).max()
            assert rel(Po.nu[i], g["nu"][i]) < 1e-10


@pytest.mark.parametrize("name", NAMES)
def test_guided_solve_and_llikelihood_two_restatements_agree(name):
    """solve!(Euler(), X, x0, W, Po) + llikelihood(LeftRule(), X, Po) on the same driving Wiener path"""
    c = case(name)
    Po = build(c)
    ref = c.oracle_proposal()
    for p, skip in ((0, 0), (1, 0), (2, 3)):
        W = o.wiener_sample(c.tt, c.mp, 99, p, 0)
        Xo = o.solve_guided(ref, c.x0, W)
        Xn = jr.solve_euler(c.x0, W, Po)
        # the guide is stiff towards T (M ~ 1e10): rounding differences of the two inverses are amplified along the path;
        # 1e-9 relative on paths, 1e-8 on ll is the same tolerance the GPU tests state for non-polynomial drifts
        assert rel(Xn, Xo) < 1e-9, (name, rel(Xn, Xo))
        llo, lln = o.llikelihood(ref, Xo, skip=skip), jr.llikelihood(Xo, Po, skip=skip)
        assert abs(lln - llo) <= 1e-8 * (1 + abs(llo)), (name, llo, lln)


def test_fhn_path_against_50_digit_arithmetic():
    """One FitzHugh-Nagumo PartialBridge path, "extreme" endpoint, evaluated a third time with mpmath at 50 digits
    (guide ODE, Euler recurrence, log-likelihood): bounds the fp64 rounding error of the oracle's path near T, where
    M_i = inv(M+_i) reaches 1e10 and the drift is stiff."""
    mp = pytest.importorskip("mpmath")
    mp.mp.dps = 50
    c = case("fhn_partialbridge_extreme", 121)
    eps, s, gam, beta, sig = (mp.mpf(x) for x in c.par)
    v = mp.mpf(c.v[0])
    tt = [mp.mpf(float(t)) for t in c.tt]
    B = mp.matrix([[1 / eps - 3 * v ** 2 / eps, -1 / eps], [gam, -1]])
    bet = mp.matrix([s / eps + 2 * v ** 3 / eps, beta])
    sg = mp.matrix([0, sig])

    def r3(f, t, y, dt):
        k1 = f(t, y)
        k2 = f(t + dt / 2, y + dt / 2 * k1)
        k3 = f(t + 3 * dt / 4, y + 3 * dt / 4 * k2)
        return y + dt * (mp.mpf(2) / 9 * k1 + mp.mpf(1) / 3 * k2 + mp.mpf(4) / 9 * k3)

    Nn = len(tt)
    L = mp.matrix([[1, 0]])
    Mp = mp.matrix([[mp.mpf(c.Sigma[0][0])]])
    mu = mp.matrix([0])
    Lt, Mt, mut = [None] * Nn, [None] * Nn, [None] * Nn
    for i in range(Nn - 2, -1, -1):
        dt = tt[i] - tt[i + 1]
        L = r3(lambda t, y: -y * B, tt[i + 1], L, dt)
        Ls = (L * sg)[0]
        Mp = r3(lambda t, y: mp.matrix([[-Ls * Ls]]), tt[i + 1], Mp, dt)
        mu = r3(lambda t, y: -L * bet, tt[i + 1], mu, dt)
        Lt[i], Mt[i], mut[i] = L, 1 / Mp[0], mu[0]
    W = o.wiener_sample(c.tt, 1, 7, 0, 0)
    y = mp.matrix([mp.mpf(float(c.x0[0])), mp.mpf(float(c.x0[1]))])
    X = np.zeros((Nn, 2))
    ll = mp.mpf(0)
    a22 = sig * sig
    for i in range(Nn - 1):
        X[i] = [float(y[0]), float(y[1])]
        q = v - mut[i] - (Lt[i] * y)[0]
        r = mp.matrix([Lt[i][0] * Mt[i] * q, Lt[i][1] * Mt[i] * q])
        bT = mp.matrix([(y[0] - y[1] - y[0] ** 3 + s) / eps, gam * y[0] - y[1] + beta])
        bA = B * y + bet
        dt = tt[i + 1] - tt[i]
        ll += ((bT[0] - bA[0]) * r[0] + (bT[1] - bA[1]) * r[1]) * dt
        dw = mp.mpf(float(W[i + 1, 0])) - mp.mpf(float(W[i, 0]))
        y = mp.matrix([y[0] + bT[0] * dt, y[1] + (bT[1] + a22 * r[1]) * dt + sig * dw])
    X[Nn - 1] = [float(y[0]), float(y[1])]
    ref = c.oracle_proposal()
    Xo = o.solve_guided(ref, c.x0, W)
    llo = o.llikelihood(ref, Xo)
    # fp64 against exact arithmetic on the same recurrence: the path is pulled onto v at T with gain ~1e10*dt, which
    # damps rather than amplifies rounding; the observed error is ~1e-12, the stated bound 1e-9 (paths) / 1e-8 (ll)
    err = np.abs(Xo - X).max()
    assert err < 1e-9 * (1 + np.abs(X).max()), err
    assert abs(llo - float(ll)) < 1e-8 * (1 + abs(float(ll))), (llo, float(ll))


# --------------------------------------------------------------------------- round 3: the application loop (f1 / f2 twins)
class _SpecNoise:
    """The noise of the specification (bhip-philox-v3) handed to the second restatement: Wiener paths of segment i at iteration
    `it` (stream 0, pairs offset by i*2^24 = normals offset by i*2^25), the normals of rand(pi0) (stream 2), the uniform
    (stream 1).  Only the generator comes from the oracle library; every use of the numbers is the restatement's own."""

    def __init__(self, tts, mp, d, seed, path):
        self.tts, self.mp, self.d, self.seed, self.path = tts, mp, d, seed, path

    def wiener(self, i, it):
        tt = self.tts[i]
        N = len(tt)
        z = o.normals(self.seed, self.path, it, i << 25, (N - 1) * self.mp).reshape(N - 1, self.mp)
        W = np.zeros((N, self.mp))
        for j in range(1, N):
            W[j] = W[j - 1] + np.sqrt(tt[j] - tt[j - 1]) * z[j - 1]
        return W

    def randn(self, it):
        import ctypes as C
        lib = o.lib()
        out, pr = np.zeros(self.d + 1), (C.c_double * 2)()
        for k in range(0, self.d, 2):
            lib.bo_normal_pair_stream(C.c_uint64(self.seed), C.c_uint32(self.path), C.c_uint32(2), C.c_uint32(it), C.c_uint32(k >> 1), pr)
            out[k], out[k + 1] = pr[0], pr[1]
        return out[:self.d]

    def rand(self, it):
        return o.uniform_accept(self.seed, self.path, it)


def test_smoothing_loop_with_shared_guides_second_restatement():
    """f1: chained GuidedBridge segments (LinPro-2 target, a different LinPro auxiliary), gpupdate links, pCN on the start, joint
    accept, mcnext! -- bo_smooth_mcmc against the numpy restatement of smoothing.jl:99-213, on the same specification noise"""
    rng = np.random.default_rng(5)
    m, M, d = 3, 40, 2
    B = np.array([[-1, 0.1], [-0.2, -1]])
    sig = 2 * np.array([[-0.212887, 0.0687025], [0.193157, 0.388997]])
    L, Sig = np.array([[1.0, 0.0]]), np.array([[0.05]])
    tgrid = np.linspace(0, 0.3 * m, m * M + 1)
    obs = rng.standard_normal((m + 1, 1))
    P, Pt = jr.LinPro(B, [0.02, 0.03], sig), jr.LinPro(0.8 * B, [0.0, 0.0], sig)
    par, apar = o.linpro_par(B, [0.02, 0.03], sig), o.linpro_par(0.8 * B, [0.0, 0.0], sig)
    # both sides build their own chain of proposals backwards (test/smoothing.jl:73-85)
    H, v = jr.gpupdate(np.diag([np.inf] * d), np.zeros(d), np.eye(d), 0.5 * np.eye(d), np.array([obs[m, 0], 0.0]))
    Ho, vo = o.gpupdate(np.diag([np.inf] * d), np.zeros(d), np.eye(d), 0.5 * np.eye(d), np.array([obs[m, 0], 0.0]))
    assert np.allclose(H, Ho, rtol=1e-12) and np.allclose(v, vo, rtol=1e-12)
    tts, Po, refs = [None] * m, [None] * m, [None] * m
    for i in range(m - 1, -1, -1):
        tts[i] = tgrid[i * M:(i + 1) * M + 1].copy()
        Po[i] = jr.GuidedBridge(tts[i], P, Pt, v, H)
        Hd, V = o.gp_hv(tts[i], d, d, o.AUX_LINPRO, apar, vo, Ho)
        refs[i] = o.proposal_hv(tts[i], d, d, o.MODEL_LINPRO, par, o.AUX_LINPRO, apar, Hd, V)
        H, v = jr.gpupdate(Po[i].Hd[0], Po[i].V[0], L, Sig, obs[i])
        Ho, vo = o.gpupdate(Hd[0], V[0], L, Sig, obs[i])
        assert np.allclose(H, Ho, rtol=1e-10) and np.allclose(v, vo, rtol=1e-10)
    iters = 25
    w_new = np.sqrt(rng.uniform(0.05, 0.5, iters)); w_old = np.sqrt(1 - w_new ** 2)
    seed, path = 17, 41
    ro = o.smooth_mcmc(refs, vo, o.chol_lower(Ho), w_old, w_new, seed, path, stats=True)
    rj = jr.smooth((v, H), tts, P, Po, jr.llikelihood, _SpecNoise(tts, d, d, seed, path), iters, w_new, w_old)
    assert rj["acc"] == ro["acc"] and 0 < ro["acc"] < iters
    for k, tol in (("X", 1e-9), ("W", 1e-12), ("y0", 1e-9), ("mean", 1e-9), ("m2", 1e-8), ("ll", 1e-8)):
        assert np.abs(rj[k] - ro[k]).max() <= tol * (1 + np.abs(ro[k]).max()), k


def test_adaptive_smoothing_loop_second_restatement():
    """f2: Lorenz, LinearAppr auxiliaries along a path, the index-based Heun guide, gpupdate, and the adaptation of
    smoothing.jl:130-160 (re-linearisation around the chain's mcnext! means, new pi0, newblock, doaccept) -- bo_smooth_adaptive
    (and its pieces bo_linearappr, bo_gp_hv_heuni, bo_gpupdate, bo_mcnext) against the numpy restatement, same noise"""
    m, M = 2, 30
    par = [10.0, 20.0, 8 / 3, 3.0, 3.0, 3.0]
    P = jr.Lorenz(par[:3], par[3:])
    tgrid = np.linspace(0.0, 0.12, m * M + 1)
    Yall = np.stack([1.5 + tgrid, -1.5 + 2 * tgrid, 25.0 - tgrid], 1)
    tts = [tgrid[i * M:(i + 1) * M + 1].copy() for i in range(m)]
    Y0 = [Yall[i * M:(i + 1) * M + 1] for i in range(m)]
    L, Sig = np.eye(3), 0.5 * np.eye(3)
    obs = Yall[::M] + 0.3
    HT, vT = jr.gpupdate(1e3 * np.eye(3), np.zeros(3), L, Sig, obs[m])
    HTo, vTo = o.gpupdate(1e3 * np.eye(3), np.zeros(3), L, Sig, obs[m])
    assert np.allclose(HT, HTo, rtol=1e-12) and np.allclose(vT, vTo, rtol=1e-12)
    # the pieces: linearappr and the index-based Heun guide of one segment
    Bo, bo_, So = o.linearappr(o.MODEL_LORENZ, 3, 3, par, tts[1], Y0[1])
    la = jr.LinearAppr(tts[1], Y0[1], P)
    assert np.allclose(np.stack(la.Bs), Bo, rtol=1e-14) and np.allclose(np.stack(la.bs), bo_, rtol=1e-14)
    Hdo, Vo = o.gp_hv_heuni(tts[1], 3, 3, Y0[1], Bo, bo_, So, vTo, HTo)
    gb = jr.GuidedBridgeLA(tts[1], P, la, vT, HT)
    assert np.abs(np.stack(gb.Hd) - Hdo).max() <= 1e-11 * np.abs(Hdo).max() and np.abs(np.stack(gb.V) - Vo).max() <= 1e-11 * np.abs(Vo).max()
    # the loop, through two adaptations
    iters, adaptit = 14, 5
    rng = np.random.default_rng(3)
    w_new = np.sqrt(rng.uniform(0.05, 0.4, iters)); w_old = np.sqrt(1 - w_new ** 2)
    seed, path = 7, 3
    ro = o.smooth_adaptive(o.MODEL_LORENZ, 3, 3, par, np.stack(tts), np.stack(Y0), L, Sig, obs[:m], HTo, vTo, w_old, w_new, adaptit, 10 ** 6, seed, path)
    H, v, Po = HT, vT, [None] * m
    for i in range(m - 1, -1, -1):
        Po[i] = jr.GuidedBridgeLA(tts[i], P, jr.LinearAppr(tts[i], Y0[i], P), v, H)
        H, v = jr.gpupdate(Po[i].Hd[0], Po[i].V[0], L, Sig, obs[i])
    rj = jr.smooth((v, H), tts, P, Po, jr.llikelihood_indexed, _SpecNoise(tts, 3, 3, seed, path), iters, w_new, w_old,
                   L=L, Sigma=Sig, obs=obs, HT=HT, vT=vT, adaptit=adaptit, adaptmax=10 ** 6)
    assert rj["acc"] == ro["acc"] and ro["acc"] >= 2
    for k, tol in (("X", 1e-9), ("W", 1e-12), ("y0", 1e-9), ("mean", 1e-9), ("m2", 1e-8), ("ll", 1e-8), ("mu", 1e-9), ("H", 1e-9)):
        assert np.abs(rj[k] - ro[k]).max() <= tol * (1 + np.abs(ro[k]).max()), k
    assert np.abs(np.stack([np.stack(p.Hd) for p in rj["Po"]]) - ro["Hd"]).max() <= 1e-9 * np.abs(ro["Hd"]).max()
    # with the moving-average variant of the re-linearisation (smoothmean, hwindow)
    ro = o.smooth_adaptive(o.MODEL_LORENZ, 3, 3, par, np.stack(tts), np.stack(Y0), L, Sig, obs[:m], HTo, vTo, w_old, w_new, adaptit, 10 ** 6, seed, path, hwindow=4)
    H, v, Po = HT, vT, [None] * m
    for i in range(m - 1, -1, -1):
        Po[i] = jr.GuidedBridgeLA(tts[i], P, jr.LinearAppr(tts[i], Y0[i], P), v, H)
        H, v = jr.gpupdate(Po[i].Hd[0], Po[i].V[0], L, Sig, obs[i])
    rj = jr.smooth((v, H), tts, P, Po, jr.llikelihood_indexed, _SpecNoise(tts, 3, 3, seed, path), iters, w_new, w_old,
                   L=L, Sigma=Sig, obs=obs, HT=HT, vT=vT, adaptit=adaptit, adaptmax=10 ** 6, smoothmean=True, hwindow=4)
    assert rj["acc"] == ro["acc"]
    for k in ("X", "mean", "mu"):
        assert np.abs(rj[k] - ro[k]).max() <= 1e-9 * (1 + np.abs(ro[k]).max()), k
