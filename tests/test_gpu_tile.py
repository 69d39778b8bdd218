"""GPU parity tests of the large-state-dimension path (config C5: LinPro d = 32 guided bridge on the
fp64 MFMA tile kernel, bhip_tile_kernel.h), through the C ABI, against the CPU oracle.

Tolerance (fp64): the oracle follows the reference literally -- an LU solve Hdiamond_i \\ (V_i - x) per
step and unfused left-to-right dot products (src/guip.jl:192-193) -- while the device multiplies by the
pre-inverted matrix and accumulates on the matrix cores with fused multiply-adds.  Stated tolerance:
|X - X_oracle| <= 1e-9 * (1 + max|X|), |ll - ll_oracle| <= 1e-8 * (1 + |ll|) (SURVEY 7 "hard parts").
The Wiener paths themselves are bit-exact (same generator, same operation order).
"""
import math

import numpy as np
import pytest
import torch

import bridgehip as bh
import oracle as o
import problems

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    return bh.default_context(0)


def _close(X, Xr, ll, llr):
    assert np.abs(X - Xr).max() <= 1e-9 * (1 + np.abs(Xr).max()), np.abs(X - Xr).max()
    assert np.all(np.abs(ll - llr) <= 1e-8 * (1 + np.abs(llr))), np.abs(ll - llr).max()


@pytest.mark.parametrize("kind", [o.GUIDE_HV, o.GUIDE_NUH], ids=["guidedbridge", "nuh"])
@pytest.mark.parametrize("d", [32, 16])
def test_tile_kernel_external_W_vs_oracle(ctx, d, kind):
    c = problems.linpro_big_case(d, 151, kind)
    P = 40                                     # neither a multiple of 64 nor of 16: tail lanes
    Po, ref = c.bh_proposal(bh, ctx), c.oracle_proposal()
    Wh = np.stack([o.wiener_sample(c.tt, d, 11, p, 0) for p in range(P)])
    W = bh.EnsemblePath.from_paths(c.tt, Wh, ctx)
    ll = ctx.empty(P)
    X = bh.solve(bh.Euler(), c.x0, W, Po, ll=ll)
    Xh = X.paths()
    Xr = np.stack([o.solve_guided(ref, c.x0, Wh[p]) for p in range(P)])
    llr = np.array([o.llikelihood(ref, Xr[p]) for p in range(P)])
    _close(Xh, Xr, ll.cpu().numpy(), llr)
    assert np.array_equal(Xh[:, 0, :], np.tile(c.x0, (P, 1)))
    if kind == o.GUIDE_HV:
        assert np.array_equal(Xh[:, -1, :], np.tile(c.v, (P, 1)))       # endpoint rule src/euler.jl:241-242
    ll7 = ctx.empty(P)
    bh.solve(bh.Euler(), c.x0, W, Po, ll=ll7, skip=7)
    ref7 = np.array([o.llikelihood(ref, Xr[p], skip=7) for p in range(P)])
    assert np.all(np.abs(ll7.cpu().numpy() - ref7) <= 1e-8 * (1 + np.abs(ref7)))


def test_tile_kernel_fused_noise(ctx):
    d, P = 32, 200
    c = problems.linpro_big_case(d, 101)
    Po, ref = c.bh_proposal(bh, ctx), c.oracle_proposal()
    X, W, ll = bh.sample_solve(c.x0, Po, P, seed=5, iter=3, path0=1000, store_W=True)
    Wh = W.paths()
    for p in (0, 15, 16, 63, 64, 199):          # Wiener paths are bit-exact (Philox pair exchange between lanes)
        assert np.array_equal(Wh[p], o.wiener_sample(c.tt, d, 5, 1000 + p, 3))
    ll2 = ctx.empty(P)
    X2 = bh.solve(bh.Euler(), c.x0, W, Po, ll=ll2)
    assert torch.equal(X2.data, X.data) and torch.equal(ll2, ll)      # fused == separate passes, same arithmetic
    Xh, llh = X.paths(), ll.cpu().numpy()
    for p in (0, 77, 199):
        Xr = o.solve_guided(ref, c.x0, Wh[p])
        _close(Xh[p], Xr, llh[p:p + 1], np.array([o.llikelihood(ref, Xr)]))
    # sharding invariance and ll-only mode
    Xb, _, llb = bh.sample_solve(c.x0, Po, 100, seed=5, iter=3, path0=1100)
    assert torch.equal(Xb.data, X.data[:, :, 100:]) and torch.equal(llb, ll[100:])
    _, _, ll3 = bh.sample_solve(c.x0, Po, P, seed=5, iter=3, path0=1000, store_X=False)
    assert torch.equal(ll3, ll)


def test_partial_bridge_large_d_maps_onto_tile_kernel(ctx):
    """PartialBridge (L,M,mu) at d = 32, two observed components: r = L'M(v - mu - Lx) is evaluated as
    (L'ML)(nu - x) with nu = L'(LL')^-1 (v - mu); equal to the oracle's literal formula within tolerance"""
    c = problems.linpro_big_case(32, 101)
    L = np.zeros((2, 32))
    L[0, 0] = 1.0
    L[1, 3], L[1, 4] = 0.5, 0.5
    v, Sig = [0.3, -0.2], 0.01 * np.eye(2)
    Po = bh.PartialBridge(c.tt, c.bh_process(bh), c.bh_aux(bh), L, v, Sig, ctx=ctx)
    Lt, Mt, mut = o.partialbridge_ode(c.tt, 32, 32, 2, c.aux, c.apar, L, Sig)
    assert np.array_equal(Po.L, Lt) and np.array_equal(Po.M, Mt) and np.array_equal(Po.mu, mut)
    ref = o.proposal_lmmu(c.tt, 32, 32, 2, c.model, c.par, c.aux, c.apar, Lt, Mt, mut, v)
    P = 24
    X, W, ll = bh.sample_solve(c.x0, Po, P, seed=2, store_W=True)
    Xh, Wh, llh = X.paths(), W.paths(), ll.cpu().numpy()
    for p in (0, 17, 23):
        Xr = o.solve_guided(ref, c.x0, Wh[p])
        _close(Xh[p], Xr, llh[p:p + 1], np.array([o.llikelihood(ref, Xr)]))
    assert np.abs(Xh[:, -1] @ L.T - np.array(v)).max() < 0.5          # pulled towards the observation


def test_large_d_unsupported_combinations_fail_loudly(ctx):
    c = problems.linpro_big_case(32, 51)
    P8 = bh.LinPro(-np.eye(8), np.zeros(8), np.eye(8))
    with pytest.raises(bh.BridgeError, match="large-d"):         # only LinPro targets run on the tile kernel
        bh.GuidedBridge(c.tt, bh.Wiener(8), P8, np.ones(8), ctx=ctx)


@pytest.mark.parametrize("d", [16, 32])
def test_tile_kernel_forward_euler_maruyama(ctx, d):
    """plain solve(EulerMaruyama(), u, W, P) of a LinPro{SVector{d}} (src/euler.jl:135-152): the tile kernel with
    a zero guide; external W against the oracle, fused noise against the external-W run."""
    c = problems.linpro_big_case(d, 121)
    P = 70
    proc = bh.PlainProcess(c.tt, c.bh_process(bh), ctx=ctx)
    x0 = 0.3 * np.ones(d)
    Wh = np.stack([o.wiener_sample(c.tt, d, 12, p, 0) for p in range(P)])
    W = bh.EnsemblePath.from_paths(c.tt, Wh, ctx)
    Xh = bh.solve(bh.EulerMaruyama(), x0, W, proc).paths()
    Xr = np.stack([o.solve_em(o.MODEL_LINPRO, d, d, c.par, c.tt, x0, Wh[p]) for p in range(P)])
    assert np.abs(Xh - Xr).max() <= 1e-9 * (1 + np.abs(Xr).max())
    assert np.array_equal(Xh[:, 0, :], np.tile(x0, (P, 1)))
    X2, W2, ll = bh.sample_solve(x0, proc, P, seed=12, store_W=True)
    assert ll is None and np.array_equal(W2.paths(), Wh)
    assert np.abs(X2.paths() - Xr).max() <= 1e-9 * (1 + np.abs(Xr).max())
    with pytest.raises(bh.BridgeError):          # llikelihood needs a guided proposal
        bh.solve(bh.EulerMaruyama(), x0, W, proc, ll=ctx.empty(P))


@pytest.mark.parametrize("d,kind", [(32, o.GUIDE_HV), (16, o.GUIDE_NUH)], ids=["d32_guidedbridge", "d16_nuh"])
def test_tile_kernel_pcn_chains(ctx, d, kind):
    """pCN Metropolis-Hastings chains at large d (SURVEY 8(d) mode M, C5 column): same slot/parity scheme as the
    path-per-lane kernel.  The Wiener state is bit-exact as long as the accept decisions agree; X and ll follow
    the stated MFMA tolerance."""
    c = problems.linpro_big_case(d, 81, kind)
    Po, ref = c.bh_proposal(bh, ctx), c.oracle_proposal()
    n, iters, rho = 40, 6, 0.95
    ch = bh.Chains(Po, c.x0, n, seed=9, path0=50)
    ll0 = ch.ll()
    r0 = o.mcmc(ref, c.x0, rho, 0, 9, 50 + 3)
    assert abs(ll0[3] - r0["ll"]) <= 1e-8 * (1 + abs(r0["ll"]))
    ch.step(rho, iters)
    acc, ll = ch.acc(), ch.ll()
    assert 0 < acc.sum() < n * iters
    X, W = ch.paths(0, n)
    for p in (0, 15, 16, 39):
        r = o.mcmc(ref, c.x0, rho, iters, 9, 50 + p)
        assert acc[p] == r["acc"]
        assert np.array_equal(W[p], r["W"])                                  # same accept history -> same W, bit for bit
        _close(X[p], r["X"], np.array([ll[p]]), np.array([r["ll"]]))
    # the statistics block and the re-materialised current X
    st = ch.stats().cpu().numpy()
    assert st[0] == n and st[1] == iters and st[2] == acc.sum() and st[5] == ll.min() and st[6] == ll.max()
    Xc = ch.current_X()
    assert np.array_equal(Xc.paths(5, 1)[0], X[5])
    llc = bh.llikelihood(bh.LeftRule(), Xc, Po).cpu().numpy()                 # stand-alone llikelihood on the tile kernel
    assert np.all(np.abs(llc - ll) <= 1e-8 * (1 + np.abs(ll)))
    # the proposal buffer of the last iteration holds solve!(Xo, Wo): finite, starts at x0
    Xo = ch.proposal_X()
    assert bool(torch.isfinite(Xo).all()) and bool((Xo[0] == torch.as_tensor(c.x0, device=Xo.device)[:, None]).all())
    nn, mean, m2 = ch.pathstats()
    Xall = X
    assert nn == n and np.abs(mean - Xall.mean(0)).max() < 1e-12
    assert np.abs(m2[40] - (Xall[:, 40] - mean[40]).T @ (Xall[:, 40] - mean[40])).max() < 1e-10


@pytest.fixture
def padded_tile(ctx):
    """BHIP_OPT_MID_VALU = 0 for the duration of a test: dimensions 4..12 on the zero-padded MFMA tile kernel instead of one path per lane"""
    ctx.set_option(bh.OPT_MID_VALU, 0)
    yield ctx
    ctx.set_option(bh.OPT_MID_VALU, 1)


@pytest.mark.parametrize("d", [4, 5, 6, 7, 10, 12, 15, 17, 24, 30, 31])
def test_tile_kernel_other_dimensions_run_zero_padded(padded_tile, d):
    """LinPro targets of any dimension 4..31 -- odd ones too since round 3 -- run on the 16- or 32-component instantiation with
    zero padding: the noise keeps the d-component counter layout (normal i*d + row of Philox call (i*d + row) >> 2: Wiener paths
    bit-exact), the ensembles hold d rows, results agree with the oracle to the MFMA tolerance.  (Dimensions 4..12 run one path per
    lane by default -- test_dimensions_4_to_12_run_one_path_per_lane -- and are sent to the tile kernel here by BHIP_OPT_MID_VALU = 0.)"""
    ctx = padded_tile
    c = problems.linpro_big_case(d, 81)
    P = 40
    Po, ref = c.bh_proposal(bh, ctx), c.oracle_proposal()
    X, W, ll = bh.sample_solve(c.x0, Po, P, seed=6, iter=2, path0=100, store_W=True)
    assert X.data.shape[1] == d and W.data.shape[1] == d
    Xh, Wh, llh = X.paths(), W.paths(), ll.cpu().numpy()
    for p in (0, 15, 16, 39):
        Wr = o.wiener_sample(c.tt, d, 6, 100 + p, 2)
        assert np.array_equal(Wh[p], Wr), (d, p)
        Xr = o.solve_guided(ref, c.x0, Wr)
        _close(Xh[p], Xr, llh[p:p + 1], np.array([o.llikelihood(ref, Xr)]))
    assert np.array_equal(Xh[:, -1, :], np.tile(c.v, (P, 1)))                     # endpoint rule, d components
    # external W, stand-alone llikelihood, plain Euler-Maruyama, pCN chains: the same padded kernel in its other modes
    ll2 = ctx.empty(P)
    X2 = bh.solve(bh.Euler(), c.x0, W, Po, ll=ll2)
    assert torch.equal(X2.data, X.data) and torch.equal(ll2, ll)
    ll3 = bh.llikelihood(bh.LeftRule(), X, Po)
    assert float((ll3 - ll).abs().max()) <= 1e-9 * (1 + float(ll.abs().max()))
    proc = bh.PlainProcess(c.tt, c.bh_process(bh), ctx=ctx)
    Xf = bh.solve(bh.EulerMaruyama(), 0.2 * np.ones(d), W, proc).paths()
    Xfr = o.solve_em(o.MODEL_LINPRO, d, d, c.par, c.tt, 0.2 * np.ones(d), Wh[7])
    assert np.abs(Xf[7] - Xfr).max() <= 1e-9 * (1 + np.abs(Xfr).max())
    ch = bh.Chains(Po, c.x0, P, seed=8)
    ch.step(0.9, 4)
    r = o.mcmc(ref, c.x0, 0.9, 4, 8, 17)
    Xc, Wc = ch.paths(17, 1)
    assert ch.acc()[17] == r["acc"] and np.array_equal(Wc[0], r["W"])
    assert abs(ch.ll()[17] - r["ll"]) <= 1e-8 * (1 + abs(r["ll"]))


def test_tile_kernel_refuses_too_large_dimensions(ctx):
    for d in (33, 40):
        B, sig = -np.eye(d), 0.5 * np.eye(d)
        with pytest.raises(bh.BridgeError, match="dimension 4 <= d <= 32"):
            bh.GuidedBridge(np.linspace(0, 1, 11), bh.LinPro(B, np.zeros(d), sig), bh.LinPro(B, np.zeros(d), sig), np.zeros(d), ctx=ctx)


def test_tile_kernel_per_path_starting_points(ctx):
    """x0_dev at large d (segment chaining: solve! returns the end point, src/euler.jl:267): every path from its own start"""
    d, P = 16, 50
    c = problems.linpro_big_case(d, 81)
    Po, ref = c.bh_proposal(bh, ctx), c.oracle_proposal()
    rng = np.random.default_rng(3)
    starts = c.x0[None, :] + 0.2 * rng.standard_normal((P, d))
    u = torch.tensor(np.ascontiguousarray(starts.T), dtype=torch.float64, device=ctx.device)      # [d][P]
    X, W, ll = bh.sample_solve(u, Po, P, seed=6, store_W=True)
    Xh, Wh, llh = X.paths(), W.paths(), ll.cpu().numpy()
    for p in (0, 17, 49):
        Xr = o.solve_guided(ref, starts[p], Wh[p])
        _close(Xh[p], Xr, llh[p:p + 1], np.array([o.llikelihood(ref, Xr)]))
        assert np.array_equal(Xh[p, 0], starts[p])


@pytest.mark.parametrize("d", [4, 5, 6, 7, 8, 9, 10, 12])
def test_dimensions_4_to_12_run_one_path_per_lane(ctx, d):
    """LinPro targets of dimension 4..8 (round 3) and 9..12 (round 4): bhip_sample_solve / bhip_solve / bhip_llikelihood run them on the path-per-lane
    kernel (k_paths<MLinPro<d>, (nu,H) form>: scalar FMAs, coefficients through the scalar unit) instead of zero padded on the
    16-row MFMA tile.  Wiener paths bit-exact vs the oracle, paths / ll at the large-d tolerance (pre-inverted guide matrix),
    agreement with the tile kernel (BHIP_OPT_MID_VALU = 0), ragged ensemble sizes, plain Euler-Maruyama, per-path starts, innovations!, pCN chains
    (slots) against the oracle, a saved state and the tile kernel's chains."""
    c = problems.linpro_big_case(d, 81)
    if d > 10:
        ctx.set_option(bh.OPT_MID_VALU, d)      # 11 and 12 are instantiated but lie beyond the default cut (10)
    try:
        _lanes_at_dimension(ctx, c, d)
    finally:
        ctx.set_option(bh.OPT_MID_VALU, 1)


def _lanes_at_dimension(ctx, c, d):
    mid = d if d > 10 else 1
    Po, ref = c.bh_proposal(bh, ctx), c.oracle_proposal()
    for P in (150, 256 + 17):
        X, W, ll = bh.sample_solve(c.x0, Po, P, seed=6, iter=2, path0=100, store_W=True)
        Xh, Wh, llh = X.paths(), W.paths(), ll.cpu().numpy()
        for p in (0, 15, 16, 63, 64, P - 1):
            Wr = o.wiener_sample(c.tt, d, 6, 100 + p, 2)
            assert np.array_equal(Wh[p], Wr), (d, p)
            Xr = o.solve_guided(ref, c.x0, Wr)
            _close(Xh[p], Xr, llh[p:p + 1], np.array([o.llikelihood(ref, Xr)]))
        assert np.array_equal(Xh[:, -1, :], np.tile(c.v, (P, 1)))
        ll2 = ctx.empty(P)
        X2 = bh.solve(bh.Euler(), c.x0, W, Po, ll=ll2)
        assert torch.equal(X2.data, X.data) and torch.equal(ll2, ll)
        ll3 = bh.llikelihood(bh.LeftRule(), X, Po)
        assert float((ll3 - ll).abs().max()) <= 1e-9 * (1 + float(ll.abs().max()))
        ctx.set_option(bh.OPT_MID_VALU, 0)
        try:
            Xp, Wp, llp = bh.sample_solve(c.x0, Po, P, seed=6, iter=2, path0=100, store_W=True)
        finally:
            ctx.set_option(bh.OPT_MID_VALU, mid)
        assert torch.equal(Wp.data, W.data)
        assert float((Xp.data - X.data).abs().max()) <= 1e-9 * (1 + float(X.data.abs().max()))
        assert float((llp - ll).abs().max()) <= 1e-8 * (1 + float(ll.abs().max()))
    proc = bh.PlainProcess(c.tt, c.bh_process(bh), ctx=ctx)
    Xf = bh.solve(bh.EulerMaruyama(), 0.2 * np.ones(d), W, proc).paths()
    Xfr = o.solve_em(o.MODEL_LINPRO, d, d, c.par, c.tt, 0.2 * np.ones(d), Wh[7])
    assert np.abs(Xf[7] - Xfr).max() <= 1e-9 * (1 + np.abs(Xfr).max())
    rng = np.random.default_rng(1)
    starts = c.x0[None, :] + 0.2 * rng.standard_normal((P, d))
    u = torch.tensor(np.ascontiguousarray(starts.T), dtype=torch.float64, device=ctx.device)
    Xs = bh.solve(bh.Euler(), u, W, Po).paths()
    for p in (0, P - 1):
        Xr = o.solve_guided(ref, starts[p], Wh[p])
        assert np.abs(Xs[p] - Xr).max() <= 1e-9 * (1 + np.abs(Xr).max()) and np.array_equal(Xs[p, 0], starts[p])
    # innovations!(EulerMaruyama, W, Y, P) (src/euler.jl:358-376): the inverse map of solve!, inv(sigma) by LU on the host; guided and plain
    Wi = bh.innovations(bh.EulerMaruyama(), X, Po).paths()
    for p in (0, 64, P - 1):
        Wo_ = o.innovations(ref, Xh[p])
        assert np.abs(Wi[p] - Wo_).max() <= 1e-9 * (1 + np.abs(Wo_).max()), (d, p)
        assert np.abs(Wi[p][:-1] - Wh[p][:-1]).max() <= 1e-7 * (1 + np.abs(Wh[p]).max())   # it IS the driving noise (the endpoint rule replaced X[N])
    Xem = bh.solve(bh.EulerMaruyama(), 0.2 * np.ones(d), W, proc)
    Wem = bh.innovations(bh.EulerMaruyama(), Xem, proc).paths()
    assert np.abs(Wem - Wh).max() <= 1e-8 * (1 + np.abs(Wh).max())
    Wem_o = o.innovations(None, Xem.paths()[7], model=o.MODEL_LINPRO, d=d, mp=d, par=c.par, tt=c.tt)
    assert np.abs(Wem[7] - Wem_o).max() <= 1e-9 * (1 + np.abs(Wem_o).max())
    # pCN chains at these dimensions run on the same kernel family (16-byte slots: current and proposal value side by side) ...
    n, iters, rho = 100, 5, 0.9
    ch = bh.Chains(Po, c.x0, n, seed=8, path0=3)
    ch.step(rho, 2)
    state = ch.save()
    ch.step(rho, iters - 2)
    acc, llc = ch.acc(), ch.ll()
    Xc, Wc = ch.paths(0, n)
    assert 0 < acc.sum() < n * iters
    for p in (0, 17, 63, 64, n - 1):
        r = o.mcmc(ref, c.x0, rho, iters, 8, 3 + p)
        assert acc[p] == r["acc"] and np.array_equal(Wc[p], r["W"]), (d, p)      # same decisions -> the same W, bit for bit
        _close(Xc[p], r["X"], np.array([llc[p]]), np.array([r["ll"]]))
    st = ch.stats().cpu().numpy()
    assert st[0] == n and st[1] == iters and st[2] == acc.sum() and st[5] == llc.min() and st[6] == llc.max()
    assert np.array_equal(ch.current_X().paths(5, 1)[0], Xc[5])                  # the re-materialised current X
    Xo = ch.proposal_X()
    assert bool(torch.isfinite(Xo).all())
    # ... resume from a saved state: the same chain
    ch2 = bh.Chains(Po, c.x0, n, seed=8, path0=3)
    ch2.load(state)
    ch2.step(rho, iters - 2)
    assert np.array_equal(ch2.acc(), acc) and np.array_equal(ch2.ll(), llc)
    # ... and agree with the zero-padded chains of the MFMA tile kernel (BHIP_OPT_MID_VALU = 0): decisions and W identical
    ctx.set_option(bh.OPT_MID_VALU, 0)
    try:
        cht = bh.Chains(Po, c.x0, n, seed=8, path0=3)
        cht.step(rho, iters)
        Xt, Wt = cht.paths(0, n)
        acct, llt = cht.acc(), cht.ll()
    finally:
        ctx.set_option(bh.OPT_MID_VALU, mid)
    assert np.array_equal(acct, acc) and np.array_equal(Wt, Wc)
    assert np.abs(Xt - Xc).max() <= 1e-9 * (1 + np.abs(Xc).max()) and np.abs(llt - llc).max() <= 1e-8 * (1 + np.abs(llc).max())


@pytest.mark.parametrize("N", [2, 3, 4, 5, 6, 9])
@pytest.mark.parametrize("d", [4, 7, 8])
def test_one_path_per_lane_short_grids_and_loop_remainders(ctx, d, N):
    """the unrolled time loop of k_paths<MLinPro<d>> (two or four steps per iteration, ragged tail with a dynamic position inside
    the Philox call) on grids shorter than / not a multiple of the unrolling, ensembles that do not fill a workgroup"""
    c = problems.linpro_big_case(d, N)
    Po, ref = c.bh_proposal(bh, ctx), c.oracle_proposal()
    X, W, ll = bh.sample_solve(c.x0, Po, 67, seed=3, iter=5, path0=9, store_W=True)
    Xh, Wh, llh = X.paths(), W.paths(), ll.cpu().numpy()
    for p in (0, 63, 66):
        Wr = o.wiener_sample(c.tt, d, 3, 9 + p, 5)
        assert np.array_equal(Wh[p], Wr), (d, N, p)
        Xr = o.solve_guided(ref, c.x0, Wr)
        _close(Xh[p], Xr, llh[p:p + 1], np.array([o.llikelihood(ref, Xr)]))
    # the chain step on the same short grids (slot prefetch clamped to the grid), with a skipped first term where there is one
    skip = 1 if N > 3 else 0
    ch = bh.Chains(Po, c.x0, 70, seed=4, path0=2, skip=skip)
    ch.step(0.8, 4)
    acc, llc = ch.acc(), ch.ll()
    Xc, Wc = ch.paths(0, 70)
    for p in (0, 63, 64, 69):
        r = o.mcmc(ref, c.x0, 0.8, 4, 4, 2 + p, skip=skip)
        assert acc[p] == r["acc"] and np.array_equal(Wc[p], r["W"]), (d, N, p)
        _close(Xc[p], r["X"], np.array([llc[p]]), np.array([r["ll"]]))


@pytest.mark.parametrize("d", [4, 5, 6, 7])
def test_mid_dimension_guidedbridge_with_diagonal_hdiamond(ctx, d):
    """Advisor r3 (high): with sigma = c*I and B~ = -I every Hdiamond_i -- and its inverse -- is exactly diagonal.  finish_guide's
    range check used to read the (nu, H) rows of the 4..8 family with the d <= 3 GUIDE_HV offsets, landed on an off-diagonal
    (exactly zero) entry of inv(Hdiamond) and refused a valid GuidedBridge with "Hdiamond singular".  Every other GPU test
    uses a dense sigma.  src/guip.jl:165-193."""
    B = -np.eye(d) + 0.1 * np.diag(np.arange(d) / d)       # diagonal target drift too
    sig = 0.5 * np.eye(d)
    par = o.linpro_par(B, np.zeros(d), sig)
    apar = o.linpro_par(-np.eye(d), np.zeros(d), sig)
    c = problems.Case(f"linpro{d}_isotropic", np.linspace(0, 1.0, 61), np.zeros(d), o.MODEL_LINPRO, par, o.AUX_LINPRO, apar,
                      o.GUIDE_HV, d, d, v=0.5 * np.ones(d), exact=False)
    ref = c.oracle_proposal()
    for mid in (1, 0):                                     # one path per lane, and the zero-padded tile kernel
        ctx.set_option(bh.OPT_MID_VALU, mid)
        try:
            Po = c.bh_proposal(bh, ctx)                    # used to raise BHIP_EUNSUPPORTED here
            X, W, ll = bh.sample_solve(c.x0, Po, 70, seed=3, iter=1, store_W=True)
        finally:
            ctx.set_option(bh.OPT_MID_VALU, 1)
        Xh, Wh, llh = X.paths(), W.paths(), ll.cpu().numpy()
        for p in (0, 33, 69):
            Xr = o.solve_guided(ref, c.x0, Wh[p])
            _close(Xh[p], Xr, llh[p:p + 1], np.array([o.llikelihood(ref, Xr)]))
    # a genuinely singular Hdiamond is still refused, with the message that names it
    sing = problems.Case(f"linpro{d}_singular", np.linspace(0, 1.0, 21), np.zeros(d), o.MODEL_LINPRO,
                         o.linpro_par(B, np.zeros(d), np.zeros((d, d))), o.AUX_LINPRO, o.linpro_par(-np.eye(d), np.zeros(d), np.zeros((d, d))),
                         o.GUIDE_HV, d, d, v=0.5 * np.ones(d), exact=False)
    with pytest.raises(Exception, match="singular"):
        sing.bh_proposal(bh, ctx)


@pytest.mark.parametrize("d", [5, 16])
def test_time_dependent_auxiliary_above_three_dimensions(ctx, d):
    """B~(t), beta~(t) are functions of t throughout the reference (src/partialbridge.jl:13-15, src/linpro.jl:188-189).  Round 5 lifts the
    "time-constant auxiliary" restriction of the large-d paths for LinPro targets: a LinearAppr auxiliary whose coefficients differ at every
    grid index (B_i, xx_i, b_i: NOT the linearisation of the target -- a genuinely different affine drift per step), GuidedBridge by the
    index-based Heun guide (src/guip.jl:181-189), at d = 5 (one path per lane: per-step B~_i, beta~_i in the coefficient rows) and d = 16
    (the tile kernel: they enter its per-step matrices A_i, b_i, c0_i).  Fresh proposals with the fused log-likelihood, the stand-alone
    llikelihood and pCN chains against the oracle at the large-d tolerance."""
    N = 81
    rng = np.random.default_rng(11)
    tt = np.linspace(0.0, 0.8, N)
    G = rng.standard_normal((d, d)) / math.sqrt(d)
    B = -np.eye(d) + 0.2 * G
    sig = 0.6 * np.eye(d) + 0.05 * rng.standard_normal((d, d)) / math.sqrt(d)
    mu = 0.1 * rng.standard_normal(d)
    x0, v = 0.2 * rng.standard_normal(d), 0.3 * np.ones(d)
    # the auxiliary by grid index
    xx = 0.3 * np.sin(np.outer(1.0 + tt, np.arange(1, d + 1)))
    Bi = np.stack([-(1.0 + 0.7 * t) * np.eye(d) + 0.1 * math.cos(3 * t) * G.T for t in tt])
    bi = np.stack([0.2 * np.cos((2.0 + k) * tt) for k in range(d)], axis=1)
    Si = np.broadcast_to(sig, (N, d, d)).copy()
    P = bh.LinPro(B, mu, sig)
    Po = bh.GuidedBridge(tt, P, bh.LinearAppr(xx, Bi, bi, Si), v, ctx=ctx)
    par = o.linpro_par(B, mu, sig)
    ref = o.proposal_hv(tt, d, d, o.MODEL_LINPRO, par, o.AUX_LINEARAPPR, o.linearappr_par(tt, xx, Bi, bi, Si), Po.Hd, Po.V)
    Hd, V = o.gp_hv_heuni(tt, d, d, xx, Bi, bi, Si, v)
    assert np.abs(Po.Hd - Hd).max() <= 1e-12 * (1 + np.abs(Hd).max()) and np.abs(Po.V - V).max() <= 1e-12 * (1 + np.abs(V).max())
    X, W, ll = bh.sample_solve(x0, Po, 70, seed=13, iter=1, path0=4, store_W=True)
    Xh, Wh, llh = X.paths(), W.paths(), ll.cpu().numpy()
    ll2 = bh.llikelihood(bh.LeftRule(), X, Po).cpu().numpy()
    for p in (0, 33, 69):
        Wr = o.wiener_sample(tt, d, 13, 4 + p, 1)
        assert np.array_equal(Wh[p], Wr)
        Xr = o.solve_guided(ref, x0, Wr)
        llr = o.llikelihood(ref, Xr)
        assert np.abs(Xh[p] - Xr).max() <= 1e-9 * (1 + np.abs(Xr).max()), (d, p)
        assert abs(llh[p] - llr) <= 1e-8 * (1 + abs(llr)) and abs(ll2[p] - llr) <= 1e-8 * (1 + abs(llr)), (d, p, llh[p], ll2[p], llr)
    assert abs(llr) > 1e-3                                   # (the auxiliary does differ from the target: the weights are not trivial)
    ch = bh.Chains(Po, x0, 40, seed=8)
    ch.step(0.9, 5)
    acc, llc = ch.acc(), ch.ll()
    same = 0
    for p in (0, 13, 39):
        r = o.mcmc(ref, x0, 0.9, 5, 8, p)
        if acc[p] == r["acc"]:
            same += 1
            assert abs(llc[p] - r["ll"]) <= 1e-8 * (1 + abs(r["ll"]))
    assert same >= 2
