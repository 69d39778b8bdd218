"""User processes with a STATE-DEPENDENT diffusion coefficient sigma(t,x,P) (bhip_model_define_sigma): the second
half of the reference's extension point "define Bridge.b and Bridge.sigma for your own type" (README.md:69-77),
with a = sigma*sigma' (src/types.jl:32) and constdiff(P) = false.

CPU part : the oracle's stand-in processes (BO_MODEL_SDIFF1/2) and its non-constant-diffusivity
           log-likelihood terms (src/partialbridge.jl:79-84) against an independent numpy evaluation;
           definition / compilation / error reporting of the texts (hipRTC compiles without a GPU).
GPU part : the texts below, compiled at run time, against the oracle bit for bit -- plain
           Euler-Maruyama, guided solves, llikelihood and pCN chains for PartialBridge; the calls the
           reference itself cannot evaluate (its other !constdiff branches name unbound variables,
           SURVEY D8) must fail loudly.
"""
import numpy as np
import pytest
import torch

import bridgehip as bh
import oracle as o
import problems

# the same expressions as oracle/bridge_oracle.c (operation order matters: results are compared bitwise)
SD1_B = "o[0] = par[0] * (par[1] - x[0]);"
SD1_S = "s[0] = par[2] * sqrt(1.0 + x[0] * x[0]);"
SD2_B = "o[0] = par[0] * (par[1] - x[0]) + par[2] * x[1];  o[1] = par[3] * (par[4] - x[1]);"
SD2_S = "s[0] = par[5] * sqrt(1.0 + x[0] * x[0]);  s[2] = par[7] * x[1];  s[3] = par[6];"
PAR1 = [1.2, 0.3, 0.5]
PAR2 = [1.1, 0.2, 0.4, 0.9, -0.1, 0.5, 0.6, 0.3]


def _case2(N=151):
    """2-d state-dependent target, first coordinate observed at T = 1 (PartialBridge), constant-sigma auxiliary"""
    th1, m1, c, th2, m2, s1, s2, s3 = PAR2
    B = np.array([[-th1, c], [0.0, -th2]])
    beta = np.array([th1 * m1, th2 * m2])
    sig = np.array([[s1, 0.1], [0.0, s2]])
    apar = np.concatenate([o.cm(B), beta, o.cm(sig)])
    return problems.Case("sdiff2_partialbridge", problems.tau_grid(1.0, N), [0.3, -0.2], o.MODEL_SDIFF2, PAR2, o.AUX_AFFINE, apar,
                         o.GUIDE_LMMU, 2, 2, m=1, L=[[1.0, 0.0]], v=[0.5], Sigma=[[1e-2]])


# --------------------------------------------------------------------------- CPU
def test_oracle_state_dependent_models_and_ll_terms():
    x = np.array([0.7, -0.4])
    th1, m1, c, th2, m2, s1, s2, s3 = PAR2
    assert np.array_equal(o.b(o.MODEL_SDIFF2, 2, PAR2, 0.0, x), [th1 * (m1 - x[0]) + c * x[1], th2 * (m2 - x[1])])
    S = np.array([[s1 * np.sqrt(1.0 + x[0] * x[0]), s3 * x[1]], [0.0, s2]])
    assert np.allclose(o.a(o.MODEL_SDIFF2, 2, 2, PAR2, 0.0, x), S @ S.T, rtol=1e-15, atol=0)
    assert np.array_equal(o.a(o.MODEL_SDIFF1, 1, 1, PAR1, 0.0, [0.7]), [[(PAR1[2] * np.sqrt(1.0 + 0.49)) ** 2]])
    # llikelihood of a PartialBridge with a non-constant a: the constdiff sum plus the two extra terms
    cse = _case2(61)
    g = cse.oracle_guide()
    P = cse.oracle_proposal(g)
    W = o.wiener_sample(cse.tt, 2, 3, 0, 0)
    X = o.solve_guided(P, cse.x0, W)
    ll = o.llikelihood(P, X)
    B, beta, sig = o.uncm(cse.apar[:4], 2, 2), cse.apar[4:6], o.uncm(cse.apar[6:], 2, 2)
    at = sig @ sig.T
    ref = 0.0
    for i in range(len(cse.tt) - 1):
        xi, dt = X[i], cse.tt[i + 1] - cse.tt[i]
        L, M, mu = g["L"][i], g["M"][i], g["mu"][i]
        r = L.T @ M @ (np.asarray(cse.v) - mu - L @ xi)
        bt = np.array([th1 * (m1 - xi[0]) + c * xi[1], th2 * (m2 - xi[1])])
        Sx = np.array([[s1 * np.sqrt(1.0 + xi[0] ** 2), s3 * xi[1]], [0.0, s2]])
        A = Sx @ Sx.T - at
        H = L.T @ M @ L
        ref += (bt - (B @ xi + beta)) @ r * dt - 0.5 * np.trace(A @ H) * dt + 0.5 * (r @ A @ r) * dt
    assert abs(ll - ref) <= 1e-11 * max(1.0, abs(ref))
    # the extra terms matter (this is not the constdiff value)
    ref_cd = sum(((np.array([th1 * (m1 - X[i][0]) + c * X[i][1], th2 * (m2 - X[i][1])]) - (B @ X[i] + beta))
                  @ (g["L"][i].T @ g["M"][i] @ (np.asarray(cse.v) - g["mu"][i] - g["L"][i] @ X[i]))) * (cse.tt[i + 1] - cse.tt[i])
                 for i in range(len(cse.tt) - 1))
    assert abs(ll - ref_cd) > 1e-6


def test_define_sigma_text_validation():
    h = bh.Context(-1)
    P = bh.UserProcess(2, SD2_B, PAR2, sigma_src=SD2_S, mp=2, ctx=h)
    assert P.model_id >= 1000 and P.mp == 2 and len(P.params()) == 8
    with pytest.raises(bh.BridgeError, match="undeclared identifier"):
        bh.UserProcess(1, SD1_B, PAR1, sigma_src="s[0] = nope;", ctx=h)
    with pytest.raises(bh.BridgeError, match="empty"):
        bh.UserProcess(1, SD1_B, PAR1, sigma_src="  ", ctx=h)
    with pytest.raises(bh.BridgeError, match="exclusive"):
        bh.UserProcess(1, SD1_B, PAR1, sigma=[[1.0]], sigma_src=SD1_S, ctx=h)
    # the guide ODEs only involve the auxiliary process: same coefficients as for any other target
    c = _case2(41)
    Po = bh.PartialBridge(c.tt, P, c.bh_aux(bh), c.L, c.v, c.Sigma, ctx=h)
    g = c.oracle_guide()
    assert np.array_equal(Po.L, g["L"]) and np.array_equal(Po.M, g["M"]) and np.array_equal(Po.mu, g["mu"])


# --------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("d", [1, 2])
def test_forward_euler_maruyama_state_dependent_sigma(d):
    ctx = bh.default_context(0)
    if d == 1:
        P, model, par, x0 = bh.UserProcess(1, SD1_B, PAR1, sigma_src=SD1_S, ctx=ctx), o.MODEL_SDIFF1, PAR1, [0.4]
    else:
        P, model, par, x0 = bh.UserProcess(2, SD2_B, PAR2, sigma_src=SD2_S, mp=2, ctx=ctx), o.MODEL_SDIFF2, PAR2, [0.3, -0.2]
    tt = np.linspace(0.0, 1.0, 202)
    npaths = 70
    proc = bh.PlainProcess(tt, P, ctx=ctx)
    X, W, _ = bh.sample_solve(x0, proc, npaths, seed=31, store_W=True)
    Xh, Wh = X.paths(), W.paths()
    for p in range(npaths):
        assert np.array_equal(Wh[p], o.wiener_sample(tt, d, 31, p, 0))
        assert np.array_equal(Xh[p], o.solve_em(model, d, d, par, tt, x0, Wh[p])), p
    assert torch.equal(bh.solve(bh.EulerMaruyama(), x0, W, proc).data, X.data)      # external W, same kernel body


@pytest.mark.gpu
def test_partial_bridge_with_state_dependent_sigma():
    ctx = bh.default_context(0)
    c = _case2()
    P = bh.UserProcess(2, SD2_B, PAR2, sigma_src=SD2_S, mp=2, ctx=ctx)
    Po = bh.PartialBridge(c.tt, P, c.bh_aux(bh), c.L, c.v, c.Sigma, ctx=ctx)
    ref = c.oracle_proposal()
    npaths = 70
    X, W, ll = bh.sample_solve(c.x0, Po, npaths, seed=32, store_W=True)
    Xh, Wh, llh = X.paths(), W.paths(), ll.cpu().numpy()
    for p in range(npaths):
        Xr = o.solve_guided(ref, c.x0, Wh[p])
        assert np.array_equal(Xh[p], Xr), p
        assert llh[p] == o.llikelihood(ref, Xr), p
    assert torch.equal(bh.llikelihood(bh.LeftRule(), X, Po), ll)
    ll5 = bh.llikelihood(bh.LeftRule(), X, Po, skip=5).cpu().numpy()
    assert ll5[7] == o.llikelihood(ref, Xh[7], skip=5)
    # the guided path ends near the observation
    assert np.abs(Xh[:, -1, 0] - 0.5).max() < 0.5
    # pCN chains on it
    ch = bh.Chains(Po, c.x0, 96, seed=33)
    ch.step(0.8, 6)
    Xc, Wc = ch.paths(40, 2)
    for k, p in enumerate((40, 41)):
        r = o.mcmc(ref, c.x0, 0.8, 6, 33, p)
        assert ch.acc()[p] == r["acc"] and ch.ll()[p] == r["ll"] and np.array_equal(Wc[k], r["W"]) and np.array_equal(Xc[k], r["X"])


@pytest.mark.gpu
def test_state_dependent_sigma_calls_the_reference_cannot_evaluate():
    ctx = bh.default_context(0)
    P = bh.UserProcess(1, SD1_B, PAR1, sigma_src=SD1_S, ctx=ctx)
    tt = problems.tau_grid(1.0, 101)
    Po = bh.GuidedBridge(tt, P, bh.LinPro([[-1.2]], [0.3], [[0.5]]), [0.6], ctx=ctx)
    # the guided SOLVE is defined (drift b + a(t,x)*r, src/guip.jl:192) and matches the oracle bit for bit
    W = bh.sample(tt, bh.Wiener(1), npaths=16, seed=34, ctx=ctx)
    X = bh.solve(bh.Euler(), [0.4], W, Po)
    Hd, V = o.gp_hv(tt, 1, 1, o.AUX_LINPRO, o.linpro_par([[-1.2]], [0.3], [[0.5]]), [0.6], None)
    ref = o.proposal_hv(tt, 1, 1, o.MODEL_SDIFF1, PAR1, o.AUX_LINPRO, o.linpro_par([[-1.2]], [0.3], [[0.5]]), Hd, V)
    assert np.array_equal(X.paths()[3], o.solve_guided(ref, [0.4], W.paths()[3]))
    # ... its llikelihood is not (src/guip.jl:439-443 names the unbound Hi): refuse instead of returning the constdiff part
    for call in (lambda: bh.llikelihood(bh.LeftRule(), X, Po), lambda: bh.sample_solve([0.4], Po, 16, seed=1),
                 lambda: bh.Chains(Po, [0.4], 16, seed=1)):
        with pytest.raises(bh.BridgeError, match="PartialBridge"):
            call()
    with pytest.raises(bh.BridgeError):
        bh.innovations(bh.EulerMaruyama(), X, bh.PlainProcess(tt, P, ctx=ctx))
