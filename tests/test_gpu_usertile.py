"""A user-defined target drift at LARGE state dimension (VERDICT r1 missing #6 / next #9): the reference's extension point
"add a method Bridge.b(t, x, P::MyProcess)" (README.md:69-77, src/types.jl:23: any ContinuousTimeProcess{SVector{d}}) for
d > 3.  bhip_model_define_components takes the drift COMPONENT-WISE as HIP C++ text and compiles it into the fp64-MFMA tile
kernel with hipRTC; checked against the oracle's stand-in for such a process, Lorenz-96 with a dense constant sigma
(BO_MODEL_LORENZ96: b_k = (x_{k+1} - x_{k-2}) x_{k-1} - x_k + F), guided by a LinPro auxiliary.

Tolerance as for the built-in large-d path (tests/test_gpu_tile.py): 1e-9 relative on paths, 1e-8 on log-likelihoods
(pre-inverted Hdiamond and MFMA accumulation against the oracle's LU solve); Wiener paths bit-exact.
"""
import numpy as np
import pytest
import torch

import bridgehip as bh
import oracle as o

L96 = "o = (x[(k+1)%d] - x[(k+d-2)%d])*x[(k+d-1)%d] - x[k] + par[0];"


def problem(d, N=121, F=2.0):
    rng = np.random.default_rng(d)
    sig = 0.4 * np.eye(d) + 0.05 * rng.standard_normal((d, d)) / np.sqrt(d)
    tt = np.linspace(0.0, 0.5, N)
    x0 = F + 0.3 * rng.standard_normal(d)
    v = F + 0.3 * rng.standard_normal(d)
    Baux = -np.eye(d)
    return tt, x0, v, sig, Baux, F


def test_component_text_is_validated_without_a_gpu():
    hctx = bh.Context(-1)
    P = bh.UserProcessComponents(16, L96, [2.0], 0.5 * np.eye(16), ctx=hctx)       # compiles (hipRTC needs no device)
    assert P.model_id >= 1000 and len(P.params()) == 1 + 256
    with pytest.raises(bh.BridgeError, match="error"):
        bh.UserProcessComponents(16, "o = undefined_symbol(x[k]);", [2.0], 0.5 * np.eye(16), ctx=hctx)
    with pytest.raises(bh.BridgeError, match="state dimension 4 <= d <= 32"):
        bh.UserProcessComponents(33, L96, [2.0], np.eye(33), ctx=hctx)
    assert bh.UserProcessComponents(7, L96, [2.0], np.eye(7), ctx=hctx).model_id >= 1000      # odd dimensions too (round 3)


@pytest.mark.gpu
@pytest.mark.parametrize("d", [16, 32, 8, 9])     # 8, 9: zero padded onto the 16-component instantiation
def test_lorenz96_guided_bridge_on_the_tile_kernel(d):
    ctx = bh.default_context(0)
    tt, x0, v, sig, Baux, F = problem(d)
    P = bh.UserProcessComponents(d, L96, [F], sig, ctx=ctx)
    Pt = bh.LinPro(Baux, F * np.ones(d), sig)
    Po = bh.GuidedBridge(tt, P, Pt, v, ctx=ctx)
    par = np.concatenate([[F], o.cm(sig)])
    apar = o.linpro_par(Baux, F * np.ones(d), sig)
    ref = o.proposal_hv(tt, d, d, o.MODEL_LORENZ96, par, o.AUX_LINPRO, apar, Po.Hd, Po.V)
    npaths = 70
    X, W, ll = bh.sample_solve(x0, Po, npaths, seed=3, store_W=True)
    Xh, Wh, llh = X.paths(), W.paths(), ll.cpu().numpy()
    for p in (0, 17, 64, 69):
        assert np.array_equal(Wh[p], o.wiener_sample(tt, d, 3, p, 0))
        Xr = o.solve_guided(ref, x0, Wh[p])
        assert np.abs(Xh[p] - Xr).max() <= 1e-9 * (1 + np.abs(Xr).max()), np.abs(Xh[p] - Xr).max()
        llr = o.llikelihood(ref, Xr)
        assert abs(llh[p] - llr) <= 1e-8 * (1 + abs(llr))
    assert np.array_equal(Xh[:, -1, :], np.tile(v, (npaths, 1)))              # endpoint rule src/euler.jl:241-242
    # the nonlinearity matters: with the quadratic terms dropped the paths differ by far more than the tolerance
    lin = bh.GuidedBridge(tt, bh.LinPro(-np.eye(d), F * np.ones(d), sig), Pt, v, ctx=ctx)
    Xl, _, _ = bh.sample_solve(x0, lin, npaths, seed=3)
    assert np.abs(Xl.paths() - Xh).max() > 1e-3
    # stand-alone llikelihood of the stored ensemble and pCN chains run on the same instantiation family
    ll2 = bh.llikelihood(bh.LeftRule(), X, Po).cpu().numpy()
    assert np.all(np.abs(ll2 - llh) <= 1e-8 * (1 + np.abs(llh)))
    ch = bh.Chains(Po, x0, 48, seed=9)
    ch.step(0.9, 4)
    acc, llc = ch.acc(), ch.ll()
    for p in (0, 47):
        r = o.mcmc(ref, x0, 0.9, 4, 9, p)
        assert abs(llc[p] - r["ll"]) <= 1e-7 * (1 + abs(r["ll"]))
    assert np.isfinite(llc).all() and 0 <= acc.sum() <= 4 * 48


@pytest.mark.gpu
def test_plain_euler_maruyama_with_a_user_drift_at_d16():
    ctx = bh.default_context(0)
    d = 16
    tt, x0, v, sig, Baux, F = problem(d, N=81)
    P = bh.UserProcessComponents(d, L96, [F], sig, ctx=ctx)
    proc = bh.PlainProcess(tt, P, ctx=ctx)
    W = bh.sample(tt, bh.Wiener(d), npaths=33, seed=2, ctx=ctx)
    X = bh.solve(bh.EulerMaruyama(), x0, W, proc)
    par = np.concatenate([[F], o.cm(sig)])
    Wh, Xh = W.paths(), X.paths()
    for p in (0, 32):
        Xr = o.solve_em(o.MODEL_LORENZ96, d, d, par, tt, x0, Wh[p])
        assert np.abs(Xh[p] - Xr).max() <= 1e-9 * (1 + np.abs(Xr).max())


@pytest.mark.gpu
@pytest.mark.parametrize("d", [4, 5, 8])
def test_component_drift_at_4_to_8_runs_one_path_per_lane(d):
    """round 3: a component-wise user drift of dimension 4..8 runs on the path-per-lane kernel family like a LinPro target of that
    dimension (k_paths<MUser>, hipRTC; the drift parameters, sigma, a and inv(sigma) streamed through the scalar unit) instead of
    zero padded on the 16-row MFMA tile: against the oracle's Lorenz-96 stand-in, and against the tile kernel (BHIP_OPT_MID_VALU = 0)
    -- proposals, stored-ensemble llikelihood, plain Euler-Maruyama, innovations!, pCN chains (slots)."""
    ctx = bh.default_context(0)
    tt, x0, v, sig, Baux, F = problem(d)
    P = bh.UserProcessComponents(d, L96, [F], sig, ctx=ctx)
    Pt = bh.LinPro(Baux, F * np.ones(d), sig)
    Po = bh.GuidedBridge(tt, P, Pt, v, ctx=ctx)
    par = np.concatenate([[F], o.cm(sig)])
    apar = o.linpro_par(Baux, F * np.ones(d), sig)
    ref = o.proposal_hv(tt, d, d, o.MODEL_LORENZ96, par, o.AUX_LINPRO, apar, Po.Hd, Po.V)
    n = 150
    X, W, ll = bh.sample_solve(x0, Po, n, seed=3, store_W=True)
    Xh, Wh, llh = X.paths(), W.paths(), ll.cpu().numpy()
    for p in (0, 63, 64, n - 1):
        assert np.array_equal(Wh[p], o.wiener_sample(tt, d, 3, p, 0))
        Xr = o.solve_guided(ref, x0, Wh[p])
        assert np.abs(Xh[p] - Xr).max() <= 1e-9 * (1 + np.abs(Xr).max()), (d, p, np.abs(Xh[p] - Xr).max())
        llr = o.llikelihood(ref, Xr)
        assert abs(llh[p] - llr) <= 1e-8 * (1 + abs(llr))
    assert np.array_equal(Xh[:, -1, :], np.tile(v, (n, 1)))
    ll2 = bh.llikelihood(bh.LeftRule(), X, Po).cpu().numpy()
    assert np.all(np.abs(ll2 - llh) <= 1e-9 * (1 + np.abs(llh)))
    ctx.set_option(bh.OPT_MID_VALU, 0)                       # the zero-padded tile kernel: same W, paths / ll to its tolerance
    try:
        Xt, Wt, llt = bh.sample_solve(x0, Po, n, seed=3, store_W=True)
        cht = bh.Chains(Po, x0, 64, seed=9)
        cht.step(0.9, 4)
        acct, llct = cht.acc(), cht.ll()
    finally:
        ctx.set_option(bh.OPT_MID_VALU, 1)
    assert np.array_equal(Wt.paths(), Wh)
    assert np.abs(Xt.paths() - Xh).max() <= 1e-9 * (1 + np.abs(Xh).max()) and np.abs(llt.cpu().numpy() - llh).max() <= 1e-8 * (1 + np.abs(llh).max())
    # plain Euler-Maruyama and its inverse map
    proc = bh.PlainProcess(tt, P, ctx=ctx)
    Xem = bh.solve(bh.EulerMaruyama(), x0, W, proc)
    Xr = o.solve_em(o.MODEL_LORENZ96, d, d, par, tt, x0, Wh[7])
    assert np.abs(Xem.paths()[7] - Xr).max() <= 1e-9 * (1 + np.abs(Xr).max())
    Wi = bh.innovations(bh.EulerMaruyama(), Xem, proc).paths()
    assert np.abs(Wi - Wh).max() <= 1e-8 * (1 + np.abs(Wh).max())
    # pCN chains on the slots: the oracle's decisions, the tile kernel's decisions
    ch = bh.Chains(Po, x0, 64, seed=9)
    ch.step(0.9, 4)
    acc, llc = ch.acc(), ch.ll()
    Xc, Wc = ch.paths(0, 64)
    for p in (0, 17, 63):
        r = o.mcmc(ref, x0, 0.9, 4, 9, p)
        assert acc[p] == r["acc"] and np.array_equal(Wc[p], r["W"]), (d, p)
        assert np.abs(Xc[p] - r["X"]).max() <= 1e-9 * (1 + np.abs(r["X"]).max()) and abs(llc[p] - r["ll"]) <= 1e-8 * (1 + abs(r["ll"]))
    assert np.array_equal(acc, acct) and np.abs(llc - llct).max() <= 1e-8 * (1 + np.abs(llc).max())


def l96_jacobian(x):
    """bderiv of the Lorenz-96 drift: d b_k / d x_j (the linearisation a caller supplies for a process of their own)."""
    d = len(x)
    J = -np.eye(d)
    for k in range(d):
        J[k, (k + 1) % d] += x[(k - 1) % d]
        J[k, (k - 2) % d] -= x[(k - 1) % d]
        J[k, (k - 1) % d] += x[(k + 1) % d] - x[(k - 2) % d]
    return J


def l96_drift(x, F):
    d = len(x)
    return np.array([(x[(k + 1) % d] - x[(k - 2) % d]) * x[(k - 1) % d] - x[k] + F for k in range(d)])


@pytest.mark.gpu
@pytest.mark.parametrize("d", [4, 5, 8, 9, 12, 16, 23, 32])
def test_component_drift_with_a_time_dependent_auxiliary(d):
    """B~(t), beta~(t) are functions of t throughout the reference (src/partialbridge.jl:13-15, src/linpro.jl:188-189); a user's process
    of dimension > 3 (README.md:69-77) with such an auxiliary was refused until round 5 at every d > 3 and until round 6 at 9 <= d <= 32 (the
    tile kernel kept -B~ constant beside a user drift).  A Lorenz-96 target with its LinearAppr linearisation along a reference curve
    (src/linpro.jl:196-204 with the caller's own bderiv), GuidedBridge by the index-based Heun guide -- fresh proposals with the fused
    log-likelihood, the stand-alone llikelihood, pCN chains -- against the oracle at the large-d tolerance:
      * one path per lane where the dimension runs there (rows carry B~_i, beta~_i per step),
      * on the MFMA tile kernel at EVERY dimension (BHIP_OPT_MID_VALU = 0 below 13): -B~_i and c_i = B~_i mu~ - beta~_i travel with the
        step row as a third per-step matrix (k_tile<.., TDA = true>)."""
    ctx = bh.Context(0)
    tt, x0, v, sig, _, F = problem(d, N=101)
    N = len(tt)
    s = (tt - tt[0]) / (tt[-1] - tt[0])
    Y = np.outer(1 - s, x0) + np.outer(s, v) + 0.2 * np.sin(np.outer(3 * s, np.arange(1, d + 1)))   # the curve linearised along
    Bi = np.stack([l96_jacobian(y) for y in Y])
    bi = np.stack([l96_drift(y, F) for y in Y])
    Si = np.broadcast_to(sig, (N, d, d)).copy()
    P = bh.UserProcessComponents(d, L96, [F], sig, ctx=ctx)
    Po = bh.GuidedBridge(tt, P, bh.LinearAppr(Y, Bi, bi, Si), v, ctx=ctx)
    par = np.concatenate([[F], o.cm(sig)])
    Hd, V = o.gp_hv_heuni(tt, d, d, Y, Bi, bi, Si, v)
    assert np.abs(Po.Hd - Hd).max() <= 1e-12 * (1 + np.abs(Hd).max()) and np.abs(Po.V - V).max() <= 1e-12 * (1 + np.abs(V).max())
    ref = o.proposal_hv(tt, d, d, o.MODEL_LORENZ96, par, o.AUX_LINEARAPPR, o.linearappr_par(tt, Y, Bi, bi, Si), Po.Hd, Po.V)
    n = 130
    Poc = bh.GuidedBridge(tt, P, bh.LinPro(-np.eye(d), F * np.ones(d), sig), v, ctx=ctx)
    for mid in ((1, 0) if d <= 12 else (1,)):       # the lanes where they run, then the tile kernel; above 12 the tile kernel anyway
        ctx.set_option(bh.OPT_MID_VALU, mid)
        X, W, ll = bh.sample_solve(x0, Po, n, seed=21, iter=2, path0=7, store_W=True)
        Xh, Wh, llh = X.paths(), W.paths(), ll.cpu().numpy()
        ll2 = bh.llikelihood(bh.LeftRule(), X, Po).cpu().numpy()
        for p in (0, 63, 64, n - 1):
            assert np.array_equal(Wh[p], o.wiener_sample(tt, d, 21, 7 + p, 2))
            Xr = o.solve_guided(ref, x0, Wh[p])
            llr = o.llikelihood(ref, Xr)
            assert np.abs(Xh[p] - Xr).max() <= 1e-9 * (1 + np.abs(Xr).max()), (d, mid, p, np.abs(Xh[p] - Xr).max())
            assert abs(llh[p] - llr) <= 1e-8 * (1 + abs(llr)) and abs(ll2[p] - llr) <= 1e-8 * (1 + abs(llr)), (d, mid, p, llh[p], ll2[p], llr)
        assert abs(llr) > 1e-3                                   # (a genuinely time-dependent auxiliary: the weights are not trivial)
        assert np.array_equal(Xh[:, -1, :], np.tile(v, (n, 1)))
        # the same paths under a time-CONSTANT auxiliary differ: the per-step coefficients are what the kernel read
        Xc = bh.solve(bh.Euler(), x0, W, Poc).paths()
        assert np.abs(Xc - Xh).max() > 1e-3
        # an external W through solve! with the fused log-likelihood == the fused proposal's
        lle = ctx.empty(n)
        Xe = bh.solve(bh.Euler(), x0, W, Po, ll=lle).paths()
        assert np.abs(Xe - Xh).max() <= 1e-9 * (1 + np.abs(Xh).max()) and np.abs(lle.cpu().numpy() - llh).max() <= 1e-8 * (1 + np.abs(llh).max())
        # pCN chains (16-byte slots on the lanes up to d = 8, the tile lines otherwise) against the oracle's decisions
        ch = bh.Chains(Po, x0, 64, seed=5)
        ch.step(0.9, 4)
        acc, llc = ch.acc(), ch.ll()
        Xk, Wk = ch.paths(0, 64)
        same = 0
        for p in (0, 17, 63):
            r = o.mcmc(ref, x0, 0.9, 4, 5, p)
            if acc[p] == r["acc"]:
                same += 1
                assert np.array_equal(Wk[p], r["W"]), (d, mid, p)
                assert np.abs(Xk[p] - r["X"]).max() <= 1e-9 * (1 + np.abs(r["X"]).max()) and abs(llc[p] - r["ll"]) <= 1e-8 * (1 + abs(r["ll"]))
        assert same >= 2
        del ch


def test_linearappr_auxiliary_for_a_user_drift_is_taken_at_every_dimension_of_the_tile_kernel():
    """(until round 6: refused above d = 8)  host context: the guide of a component-wise user drift with a LinearAppr auxiliary at d = 9
    and 32 is computed and equals the oracle's index-based Heun guide"""
    ctx = bh.Context(-1)
    for d in (9, 32):
        tt, x0, v, sig, _, F = problem(d, N=41)
        N = len(tt)
        P = bh.UserProcessComponents(d, L96, [F], sig, ctx=ctx)
        Y = np.tile(x0, (N, 1))
        Bi, bi, Si = np.stack([l96_jacobian(y) for y in Y]), np.stack([l96_drift(y, F) for y in Y]), np.broadcast_to(sig, (N, d, d)).copy()
        Po = bh.GuidedBridge(tt, P, bh.LinearAppr(Y, Bi, bi, Si), v, ctx=ctx)
        Hd, V = o.gp_hv_heuni(tt, d, d, Y, Bi, bi, Si, v)
        assert np.abs(Po.Hd - Hd).max() <= 1e-12 * (1 + np.abs(Hd).max()) and np.abs(Po.V - V).max() <= 1e-12 * (1 + np.abs(V).max())


L96_FULL = "for (int j = 0; j < d; j++) o[j] = (x[(j+1)%d] - x[(j+d-2)%d])*x[(j+d-1)%d] - x[j] + par[0];"


def test_full_form_drift_text_above_three_dimensions_is_validated_without_a_gpu():
    """bhip_model_define at 4 <= d <= 32 (round 6): the full-form body `o[0] = ...; o[1] = ...;` of README.md:69-77 is carried as a
    component-wise model; m' must equal d (constant dense sigma), a state-dependent sigma stays at d <= 3"""
    hctx = bh.Context(-1)
    P = bh.UserProcess(12, L96_FULL, [2.0], sigma=0.5 * np.eye(12), ctx=hctx)
    assert P.model_id >= 1000 and P.d == 12 and P.mp == 12
    with pytest.raises(bh.BridgeError, match="error"):
        bh.UserProcess(6, "o[0] = undefined_name;", [1.0], sigma=np.eye(6), ctx=hctx)
    with pytest.raises(bh.BridgeError, match="m' = d"):
        bh.UserProcess(6, L96_FULL, [1.0], sigma=np.ones((6, 2)), ctx=hctx)
    with pytest.raises(bh.BridgeError):
        bh.UserProcess(6, L96_FULL, [1.0], sigma_src="s[0] = 1.0;", mp=6, ctx=hctx)


@pytest.mark.gpu
@pytest.mark.parametrize("d", [5, 16, 23])
def test_full_form_drift_equals_the_component_wise_model(d):
    """the same Lorenz-96 drift given as a full-form body and component-wise: identical paths and log-likelihoods on the lanes (d = 5) and
    on the MFMA tile kernel (d = 16, zero padded d = 23), fresh proposals and pCN chains"""
    ctx = bh.Context(0)
    tt, x0, v, sig, Baux, F = problem(d, N=81)
    Pc = bh.UserProcessComponents(d, L96, [F], sig, ctx=ctx)
    Pf = bh.UserProcess(d, L96_FULL, [F], sigma=sig, ctx=ctx)
    Pt = bh.LinPro(Baux, F * np.ones(d), sig)
    Poc, Pof = bh.GuidedBridge(tt, Pc, Pt, v, ctx=ctx), bh.GuidedBridge(tt, Pf, Pt, v, ctx=ctx)
    n = 200
    Xc, _, llc = bh.sample_solve(x0, Poc, n, seed=3, iter=1)
    Xf, _, llf = bh.sample_solve(x0, Pof, n, seed=3, iter=1)
    assert np.array_equal(Xc.paths(), Xf.paths()) and torch.equal(llc, llf) and bool(torch.isfinite(llc).all())
    chc, chf = bh.Chains(Poc, x0, 128, seed=2), bh.Chains(Pof, x0, 128, seed=2)
    chc.step(0.9, 3); chf.step(0.9, 3)
    assert np.array_equal(chc.ll(), chf.ll()) and np.array_equal(chc.acc(), chf.acc())
