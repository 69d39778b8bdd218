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
