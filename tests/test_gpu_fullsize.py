"""GPU tests at BASELINE.json's full sizes (N = 1001; 65 536 / 262 144 paths), where the oracle can only
spot-check: a sample of path ids is compared bit-for-bit with the oracle, the rest is covered by
size-independent properties (checksums, endpoint rules, importance-weight unbiasedness with a much
sharper sample than the reference's m = 1000, sharding invariance, stationarity of the chains).
"""
import math

import numpy as np
import pytest
import torch

import bridgehip as bh
import oracle as o
import problems

pytestmark = pytest.mark.gpu
N = 1001


@pytest.fixture(scope="module")
def ctx():
    return bh.default_context(0)


def _case(name):
    return [c for c in problems.cases(N) if c.name == name][0]


def test_C2_ou_guided_bridge_65536_paths(ctx):
    """config C2 + K9 (test/guip.jl:245-274) with m = 65 536 instead of 1000"""
    c = _case("ou_guidedbridge")
    P = 65536
    Po = c.bh_proposal(bh, ctx)
    X, W, ll = bh.sample_solve(c.x0, Po, P, seed=2, store_W=True)
    assert torch.all(X.data[-1, 0] == c.v[0]) and torch.all(X.data[0, 0] == c.x0[0])     # pinned end, x0 stored first
    llh = ll.cpu().numpy()
    assert np.all(np.isfinite(llh))
    # importance weights: E[exp(ll)] * ptilde / p = 1 with p the OU transition density (closed form)
    beta, a, T, u, v = 0.8, math.sqrt(0.7) ** 2, 2.0, c.x0[0], c.v[0]
    K = a / (2 * beta) * (1 - math.exp(-2 * beta * T))
    lp = -0.5 * ((v - u * math.exp(-beta * T)) ** 2 / K + math.log(K) + math.log(2 * math.pi))
    w = np.exp(llh + bh.lptilde(Po, c.x0) - lp)
    stat = abs(np.mean(w - 1)) * math.sqrt(P) / np.std(w, ddof=1)
    assert stat < 4.0, stat
    assert abs(np.mean(w) - 1) < 0.02
    # spot check against the oracle, bit for bit
    ref = c.oracle_proposal()
    Xh, Wh = X.paths(0, 2), W.paths(0, 2)
    for p in (0, 1):
        Wr = o.wiener_sample(c.tt, 1, 2, p, 0)
        assert np.array_equal(Wh[p], Wr) and np.array_equal(Xh[p], o.solve_guided(ref, c.x0, Wr))
    for p in (31337, P - 1):
        Wr = o.wiener_sample(c.tt, 1, 2, p, 0)
        Xr = o.solve_guided(ref, c.x0, Wr)
        assert np.array_equal(X.paths(p, 1)[0], Xr) and llh[p] == o.llikelihood(ref, Xr)
    # checksum of checksums: llikelihood re-evaluated on the stored ensemble reproduces the fused values
    assert torch.equal(bh.llikelihood(bh.LeftRule(), X, Po), ll)


def test_C3_fhn_partial_bridge_262144_paths(ctx):
    """config C3: FitzHugh-Nagumo PartialBridge, 1001 steps, 262 144 paths"""
    c = _case("fhn_partialbridge_extreme")
    P = 262144
    Po = c.bh_proposal(bh, ctx)
    X, _, ll = bh.sample_solve(c.x0, Po, P, seed=3)
    llh = ll.cpu().numpy()
    assert np.all(np.isfinite(llh)) and bool(torch.isfinite(X.data).all())
    # the guided proposal hits the observation L x_T = v up to the observation noise scale
    end = X.data[-1, 0]
    assert float((end - c.v[0]).abs().max()) < 5e-3
    assert torch.all(X.data[0, 0] == c.x0[0]) and torch.all(X.data[0, 1] == c.x0[1])
    ref = c.oracle_proposal()
    for p in (0, 99999, P - 1):
        Xr = o.solve_guided(ref, c.x0, o.wiener_sample(c.tt, 1, 3, p, 0))
        assert np.array_equal(X.paths(p, 1)[0], Xr) and llh[p] == o.llikelihood(ref, Xr)
    # sharding invariance at full size: the second half computed as its own launch (another "GPU")
    Xb, _, llb = bh.sample_solve(c.x0, Po, P // 2, seed=3, path0=P // 2)
    assert torch.equal(Xb.data, X.data[:, :, P // 2:]) and torch.equal(llb, ll[P // 2:])
    del Xb
    # 3-d NCLAR at the same size (the "3-d" of BASELINE.json): finite, hits the observation
    c3 = _case("nclar_firstcomponent")
    Po3 = c3.bh_proposal(bh, ctx)
    X3, _, ll3 = bh.sample_solve(c3.x0, Po3, P, seed=3)
    assert bool(torch.isfinite(ll3).all()) and float((X3.data[-1, 0] - c3.v[0]).abs().max()) < 5e-3
    ref3 = c3.oracle_proposal()
    for p in (4242, P - 1):   # `==`: MNCLAR::b and the oracle share one sin (bhip_trig.h det_sin); both parts of the container
        Xr = o.solve_guided(ref3, c3.x0, o.wiener_sample(c3.tt, 1, 3, p, 0))
        assert np.array_equal(X3.paths(p, 1)[0], Xr) and float(ll3[p]) == o.llikelihood(ref3, Xr)
    # the containers of this size are kept in two buffers (EnsemblePath: 1 GiB and more), written by one launch
    assert X.nparts == 2 and X3.nparts == 2 and X.part_paths == P // 2


def test_linpro4_262144_paths_on_the_lanes(ctx):
    """the `linpro4` bench mode at its full size: LinPro d = 4 GuidedBridge (dense sigma), 1001 steps, 262 144 paths, one path per lane on
    the regrouped rows (three products per step: tolerance parity like every d > 3 path) -- finite, endpoint rule, stand-alone
    llikelihood == fused to tolerance, sharding invariance `==`, spot checks against the oracle in both parts of the container"""
    d, P = 4, 262144
    c = problems.linpro_big_case(d, N)
    Po = c.bh_proposal(bh, ctx)
    X, _, ll = bh.sample_solve(c.x0, Po, P, seed=6)
    assert X.nparts == 2 and bool(torch.isfinite(ll).all())
    assert bool((X.endpoints() == torch.as_tensor(c.v, device=ll.device)[:, None]).all())      # endpoint rule src/euler.jl:241-242
    ll2 = bh.llikelihood(bh.LeftRule(), X, Po)
    assert float((ll2 - ll).abs().max()) <= 1e-8 * (1 + float(ll.abs().max()))
    Xq, _, llq = bh.sample_solve(c.x0, Po, P // 4, seed=6, path0=3 * P // 4)
    assert torch.equal(llq, ll[3 * P // 4:]) and np.array_equal(Xq.paths(0, 64), X.paths(3 * P // 4, 64)) and np.array_equal(Xq.paths(P // 4 - 3, 3), X.paths(P - 3, 3))
    ref = c.oracle_proposal()
    for p in (0, P // 2 - 1, P // 2, P - 1):
        Xr = o.solve_guided(ref, c.x0, o.wiener_sample(c.tt, d, 6, p, 0))
        assert np.abs(X.paths(p, 1)[0] - Xr).max() <= 1e-9 * (1 + np.abs(Xr).max())
        llr = o.llikelihood(ref, Xr)
        assert abs(float(ll[p]) - llr) <= 1e-8 * (1 + abs(llr))


def _linpro_logdensity(B, mu, sig, u, T, v):
    """log transition density of dX = B(X-mu)dt + sig dW from u at 0 to v at T (src/linpro.jl:98-113)"""
    from scipy.linalg import expm, solve_continuous_lyapunov
    a = sig @ sig.T
    lam = solve_continuous_lyapunov(B, -a)
    phi = expm(T * B)
    m = phi @ (u - mu) + mu
    K = lam - phi @ lam @ phi.T
    r = v - m
    return -0.5 * (r @ np.linalg.solve(K, r) + np.linalg.slogdet(K)[1] + len(v) * math.log(2 * math.pi))


@pytest.mark.parametrize("d", [2, 3, 32])
def test_K9_importance_weights_unbiased_multivariate(ctx, d):
    """K9 (test/guip.jl:245-274) for vector LinPro targets, where the transition density is closed-form:
    mean(exp(ll) * ptilde / p) = 1.  Exercises solve! + llikelihood + lptilde jointly at distribution
    level for the path-per-lane kernel (d = 2, 3) and the MFMA tile kernel (d = 32)."""
    if d == 32:
        c = problems.linpro_big_case(32, 401)
    else:
        c = _case("linpro2_guidedbridge" if d == 2 else "linpro3_guidedbridge")
    P = 65536
    Po = c.bh_proposal(bh, ctx)
    _, _, ll = bh.sample_solve(c.x0, Po, P, seed=10 + d, store_X=False)
    llh = ll.cpu().numpy()
    B = o.uncm(c.par[:d * d], d, d)
    mu = np.asarray(c.par[d * d:d * d + d])
    sig = o.uncm(c.par[d * d + d:], d, d)
    lp = _linpro_logdensity(B, mu, sig, c.x0, c.tt[-1] - c.tt[0], np.asarray(c.v, dtype=float))
    lw = llh + bh.lptilde(Po, c.x0) - lp
    w = np.exp(lw)
    se = np.std(w, ddof=1) / math.sqrt(P)
    # discretisation bias of the Euler scheme is O(dt); allow it on top of 4 standard errors
    assert abs(np.mean(w) - 1) < 4 * se + 0.03, (np.mean(w), se)
    assert np.all(np.isfinite(llh))


def test_C4_pcn_mcmc_262144_chains(ctx):
    """config C4's per-GPU shard: 262 144 pCN chains, a few iterations"""
    c = _case("fhn_partialbridge_extreme")
    P, iters = 262144, 6
    ch = bh.Chains(c.bh_proposal(bh, ctx), c.x0, P, seed=4, path0=7 * P)       # the shard rank 7 would own
    ll0 = ch.ll()
    ch.step(0.9, iters)
    ll, acc = ch.ll(), ch.acc()
    st = ch.stats().cpu().numpy()
    assert st[0] == P and st[1] == iters and st[2] == acc.sum()
    assert abs(st[3] - ll.sum()) <= 1e-9 * np.abs(ll).sum() and st[5] == ll.min() and st[6] == ll.max()
    rate = acc.sum() / (P * iters)
    assert 0.1 < rate < 0.9, rate
    # MH targets exp(ll) * prior: accepted moves raise ll on average relative to the initial draw
    assert ll.mean() > ll0.mean()
    ref = c.oracle_proposal()
    X, W = ch.paths(1234, 2)
    for k, p in enumerate((1234, 1235)):
        r = o.mcmc(ref, c.x0, 0.9, iters, 4, 7 * P + p)
        assert acc[p] == r["acc"] and ll[p] == r["ll"] and np.array_equal(W[k], r["W"]) and np.array_equal(X[k], r["X"])


def test_C5_linpro32_guided_bridge_65536_paths(ctx):
    """config C5: LinPro d = 32 GuidedBridge, 1001 steps, 65 536 paths on the MFMA tile kernel"""
    d, P = 32, 65536
    c = problems.linpro_big_case(d, N)
    Po = c.bh_proposal(bh, ctx)
    X, _, ll = bh.sample_solve(c.x0, Po, P, seed=5)
    assert bool(torch.isfinite(ll).all())
    assert bool((X.data[0] == 0).all())                                                   # x0 = 0 stored first
    assert bool((X.data[-1] == torch.as_tensor(c.v, device=X.data.device)[:, None]).all())  # endpoint rule src/euler.jl:241-242
    # checksum of checksums: the stand-alone llikelihood of the stored ensemble reproduces the fused values
    ll2 = bh.llikelihood(bh.LeftRule(), X, Po)
    assert float((ll2 - ll).abs().max()) <= 1e-9 * (1 + float(ll.abs().max()))
    # sharding invariance: the last quarter as its own launch
    Xq, _, llq = bh.sample_solve(c.x0, Po, P // 4, seed=5, path0=3 * P // 4)
    assert torch.equal(llq, ll[3 * P // 4:]) and torch.equal(Xq.data, X.data[:, :, 3 * P // 4:])
    del Xq
    # spot check against the oracle (stated MFMA tolerance)
    ref = c.oracle_proposal()
    for p in (0, P - 1):
        Xr = o.solve_guided(ref, c.x0, o.wiener_sample(c.tt, d, 5, p, 0))
        assert np.abs(X.paths(p, 1)[0] - Xr).max() <= 1e-9 * (1 + np.abs(Xr).max())
        llr = o.llikelihood(ref, Xr)
        assert abs(float(ll[p]) - llr) <= 1e-8 * (1 + abs(llr))


def test_C4_full_run_1000_iterations_of_32768_chains(ctx):
    """config C4 as specified: one GPU's shard (32 768 chains) for the full 1000 pCN iterations, rho = 0.9;
    after 10^9 path-steps per ... chain-step the spot-checked chains still agree with the oracle bit for bit"""
    c = _case("fhn_partialbridge_extreme")
    P, iters = 32768, 1000
    ch = bh.Chains(c.bh_proposal(bh, ctx), c.x0, P, seed=4, path0=3 * P, store_X=False)      # rank 3's shard
    marks = []
    for n in (100, 400, 250, 250):
        ch.step(0.9, n)
        marks.append(ch.acc().sum())
    ll, acc = ch.ll(), ch.acc()
    assert np.all(np.isfinite(ll)) and acc.min() >= 1 and acc.max() < iters                # K10: 1 < acc < iterations
    # the acceptance rate falls during burn-in (the chains climb in ll) and then settles: the last two quarters agree
    r = [marks[0] / (P * 100), (marks[1] - marks[0]) / (P * 400), (marks[2] - marks[1]) / (P * 250), (marks[3] - marks[2]) / (P * 250)]
    assert r[0] > r[1] > r[2] - 0.01 and abs(r[2] - r[3]) < 0.02 and 0.1 < r[3] < 0.7, r
    st = ch.stats().cpu().numpy()
    assert st[1] == iters and st[2] == acc.sum()
    ref = c.oracle_proposal()
    X, W = ch.paths(777, 1)
    r = o.mcmc(ref, c.x0, 0.9, iters, 4, 3 * P + 777)
    assert acc[777] == r["acc"] and ll[777] == r["ll"] and np.array_equal(W[0], r["W"]) and np.array_equal(X[0], r["X"])
    r = o.mcmc(ref, c.x0, 0.9, iters, 4, 3 * P + P - 1)
    assert acc[P - 1] == r["acc"] and ll[P - 1] == r["ll"]


def test_K13_noise_moments_and_tails_at_scale(ctx):
    """K13 (test/wiener.jl:33-47, moments of W_T) sharpened: 2.6e8 in-kernel normals (262 144 paths x 1000 steps on a unit
    grid, so the increments ARE the normals): mean, variance, skewness, kurtosis and tail frequencies within 5 standard
    errors of the N(0,1) values, and no correlation between consecutive draws or neighbouring paths"""
    P, N = 262144, 1001
    tt = np.arange(N, dtype=np.float64)
    W = bh.sample(tt, bh.Wiener(1), npaths=P, seed=77, ctx=ctx)
    z = (W.data[1:, 0, :] - W.data[:-1, 0, :])
    n = z.numel()
    m1 = float(z.mean())
    m2 = float((z * z).mean())
    m3 = float((z ** 3).mean())
    m4 = float((z ** 4).mean())
    se = 1 / math.sqrt(n)
    assert abs(m1) < 5 * se and abs(m2 - 1) < 5 * math.sqrt(2) * se
    assert abs(m3) < 5 * math.sqrt(15) * se and abs(m4 - 3) < 5 * math.sqrt(96) * se
    for thr, p in ((1.0, 0.31731050786291415), (3.0, 0.0026997960632601866), (4.5, 6.795346249460121e-06)):
        cnt = float((z.abs() > thr).sum())
        assert abs(cnt - n * p) < 5 * math.sqrt(n * p), (thr, cnt, n * p)
    assert float(z.abs().max()) < 7.5                                                  # P(|z| > 7.5) * 2.6e8 = 1.7e-5
    # serial correlation along a path (normals 2j, 2j+1 come from one Box-Muller pair) and across neighbouring paths
    c_time = float((z[1:] * z[:-1]).mean())
    c_path = float((z[:, 1:] * z[:, :-1]).mean())
    assert abs(c_time) < 5 * se and abs(c_path) < 5 * se


def test_mcmc_reaches_the_exact_ou_bridge_law(ctx):
    """End-to-end, distribution level: the target is an Ornstein-Uhlenbeck process (beta = 0.8, a = 0.7), the auxiliary a
    DIFFERENT linear process, so proposals are wrong by the Girsanov weight and only the Metropolis-Hastings correction
    (llikelihood + accept, rows a8 + a10) makes the chains sample the true bridge.  The OU bridge given X_0 = u, X_T = v
    is Gaussian with closed-form mean and variance; 65 536 chains after 300 pCN iterations must match them at every
    grid point up to Monte-Carlo error and the O(dt) bias of the Euler scheme; the raw proposals must NOT."""
    c = _case("ou_guidedbridge")
    beta, a, T, u, v = 0.8, 0.7, 2.0, float(c.x0[0]), float(c.v[0])
    tt = c.tt
    var_t = a * (1 - np.exp(-2 * beta * tt)) / (2 * beta)                      # Var(X_t | X_0)
    var_T = var_t[-1]
    cov = np.exp(-beta * (T - tt)) * var_t                                     # Cov(X_t, X_T | X_0)
    mean_exact = u * np.exp(-beta * tt) + cov / var_T * (v - u * math.exp(-beta * T))
    var_exact = var_t - cov ** 2 / var_T
    P = 65536
    Po = c.bh_proposal(bh, ctx)
    ch = bh.Chains(Po, c.x0, P, seed=91, store_X=False)
    X0 = ch.current_X().data[:, 0, :]
    prop_mean = X0.mean(1).cpu().numpy()
    ch.step(0.7, 300)
    X = ch.current_X().data[:, 0, :]
    m, s2 = X.mean(1).cpu().numpy(), X.var(1).cpu().numpy()
    inner = slice(1, -1)
    se = np.sqrt(var_exact[inner] / P)
    # pCN chains are autocorrelated across iterations but the 65 536 chains are independent: plain standard errors apply
    assert np.abs(m[inner] - mean_exact[inner]).max() < 6 * se.max() + 4e-3, np.abs(m[inner] - mean_exact[inner]).max()
    far = (tt > 0.04) & (tt < T - 0.2)          # the Euler scheme's relative variance error grows like dt/(T-t) towards the pinned end
    assert np.abs(s2[far] / var_exact[far] - 1).max() < 0.03 and np.abs(s2 - var_exact).max() < 4e-3
    # the uncorrected proposals are visibly off (otherwise this test would not test the MH step)
    assert np.abs(prop_mean[inner] - mean_exact[inner]).max() > 5 * np.abs(m[inner] - mean_exact[inner]).max()
    acc = ch.acc().sum() / (P * 300)
    assert 0.2 < acc < 0.95


def test_C1_ou_euler_maruyama_65536_paths(ctx):
    """config C1 (README.md:69-83): OU(beta = 2, sigma = 1), Euler-Maruyama on 0:0.01:10 from x0 = 0.1 -- 65 536 paths at
    once.  The scheme's own law is known exactly: x_N = (1 - beta dt)^N x0 + noise with variance
    sigma^2 dt (1 - r^(2N)) / (1 - r^2), r = 1 - beta dt."""
    N, P = 1001, 65536
    tt = np.arange(N) * 0.01
    proc = bh.PlainProcess(tt, bh.OrnsteinUhlenbeck(2.0, 1.0), ctx=ctx)
    X, W, _ = bh.sample_solve([0.1], proc, P, seed=1, store_W=True)
    xT = X.data[-1, 0]
    r = 1 - 2.0 * 0.01
    var = 0.01 * (1 - r ** (2 * (N - 1))) / (1 - r * r)
    mean = 0.1 * r ** (N - 1)
    assert abs(float(xT.mean()) - mean) < 5 * math.sqrt(var / P)
    assert abs(float(xT.var()) / var - 1) < 5 * math.sqrt(2 / P)
    assert bool((X.data[0, 0] == 0.1).all())
    for p in (0, P - 1):
        Wr = o.wiener_sample(tt, 1, 1, p, 0)
        assert np.array_equal(W.paths(p, 1)[0], Wr)
        assert np.array_equal(X.paths(p, 1)[0], o.solve_em(o.MODEL_OU, 1, 1, [2.0, 1.0], tt, [0.1], Wr))


def test_mcmc_reaches_the_exact_partially_observed_linear_law(ctx):
    """The vector twin of the OU test: a 2-d linear target with coupled components, ONE component observed with noise at T
    (PartialBridge, L = [1 0]), an auxiliary with a different drift matrix.  X_t | X_0, L X_T + eps = v is Gaussian with
    a Kalman-type closed form (exact transition via matrix exponentials); the chains (noise dimension 2: line layout)
    must reach its mean vector and covariance matrix at every tested time, the uncorrected proposals must not."""
    from scipy.linalg import expm, solve_continuous_lyapunov
    B = np.array([[-1.0, 0.8], [-0.6, -0.7]])
    sig = np.array([[0.6, 0.0], [0.2, 0.5]])
    a = sig @ sig.T
    T, u, v, Sig = 1.5, np.array([0.4, -0.3]), np.array([0.9]), np.array([[0.01]])
    L = np.array([[1.0, 0.0]])
    tt = problems.tau_grid(T, 1001)
    P = bh.LinPro(B, np.zeros(2), sig)
    Pt = bh.LinPro(np.array([[-0.3, 0.0], [0.0, -0.3]]), np.zeros(2), sig)
    Po = bh.PartialBridge(tt, P, Pt, L, v, Sig, ctx=ctx)
    n = 65536
    ch = bh.Chains(Po, u, n, seed=92, store_X=False)
    X0 = ch.current_X().data
    ch.step(0.7, 300)
    X = ch.current_X().data                                   # [N, 2, n]
    lam = solve_continuous_lyapunov(B, -a)

    def law(t):
        Pt_, Ps_ = expm(t * B), expm((T - t) * B)
        Qt, Qs = lam - Pt_ @ lam @ Pt_.T, lam - Ps_ @ lam @ Ps_.T
        m = Pt_ @ u
        H = L @ Ps_
        S = H @ Qt @ H.T + L @ Qs @ L.T + Sig
        G = Qt @ H.T @ np.linalg.inv(S)
        return m + (G @ (v - H @ m)), Qt - G @ H @ Qt

    worst_prop = 0.0
    for i in (200, 500, 800, 950):
        mean, cov = law(tt[i])
        xi = X[i].cpu().numpy()                               # [2, n]
        se = np.sqrt(np.diag(cov) / n)
        assert np.all(np.abs(xi.mean(1) - mean) < 6 * se + 4e-3), (i, xi.mean(1), mean)
        assert np.abs(np.cov(xi) - cov).max() < 0.03 * np.abs(cov).max() + 2e-3, (i, np.cov(xi), cov)
        worst_prop = max(worst_prop, np.abs(X0[i].cpu().numpy().mean(1) - mean).max())
    assert worst_prop > 0.03                                  # the proposals alone are biased: the MH step does the work
    assert 0.15 < ch.acc().sum() / (n * 300) < 0.95


def test_mcmc_reaches_the_exact_linear_bridge_law_d32(ctx):
    """Distribution-level check of the MFMA tile kernel's chains: LinPro d = 32 target, auxiliary with another drift matrix,
    endpoint conditioned exactly (GuidedBridge).  X_t | X_0 = u, X_T = v is Gaussian in closed form; after 150 pCN
    iterations 16 384 chains must reproduce its mean vector and the diagonal of its covariance at mid-time."""
    from scipy.linalg import expm, solve_continuous_lyapunov
    d = 32
    c = problems.linpro_big_case(d, 401)
    B = o.uncm(c.par[:d * d], d, d)
    sig = o.uncm(c.par[d * d + d:], d, d)
    a = sig @ sig.T
    T, u, v = float(c.tt[-1]), np.asarray(c.x0, dtype=float), np.asarray(c.v, dtype=float)
    Po = c.bh_proposal(bh, ctx)
    n = 16384
    ch = bh.Chains(Po, u, n, seed=93, store_X=False)
    ch.step(0.8, 150)
    X = ch.current_X().data
    lam = solve_continuous_lyapunov(B, -a)
    i = 200
    t = float(c.tt[i])
    Pt_, Ps_ = expm(t * B), expm((T - t) * B)
    Qt, Qs = lam - Pt_ @ lam @ Pt_.T, lam - Ps_ @ lam @ Ps_.T
    S = Ps_ @ Qt @ Ps_.T + Qs                                  # Var(X_T | X_0)
    G = Qt @ Ps_.T @ np.linalg.inv(S)
    mean = Pt_ @ u + G @ (v - Ps_ @ Pt_ @ u)
    cov = Qt - G @ Ps_ @ Qt
    xi = X[i].cpu().numpy()                                    # [32, n]
    se = np.sqrt(np.diag(cov) / n)
    assert np.all(np.abs(xi.mean(1) - mean) < 6 * se + 5e-3), np.abs(xi.mean(1) - mean).max()
    assert np.abs(xi.var(1) / np.diag(cov) - 1).max() < 0.08
    assert 0.05 < ch.acc().sum() / (n * 150) < 0.99


def test_smoothing_loop_reaches_the_exact_linear_gaussian_smoothing_law(ctx):
    """The application loop end to end, distribution level: an Ornstein-Uhlenbeck target (beta = 0.8, a = 0.7)
    observed with noise at the four knots of three chained GuidedBridge segments whose auxiliary is a DIFFERENT linear
    process -- every proposal is wrong by its Girsanov weight, the start is drawn from the auxiliary's pi0, and only the joint
    Metropolis-Hastings decision on the summed log-likelihoods (bhip_segchains_*: pCN on y0 and on every segment's W, one
    accept) makes the chains sample the true smoothing law.  That law is Gaussian: flat prior on X(0), exact OU transitions,
    observations N(y_k; x_k, Sigma), and the N(0, piH) pseudo-prior at T the backward recursion starts from
    (supplements/smoothing/smoothing.jl:75).  32 768 chains after 300 iterations must match its mean and variance at the knots
    up to Monte-Carlo error and the O(dt) bias of the Euler scheme; the first proposals must not."""
    beta, a = 0.8, 0.7
    m, M, n, iters = 3, 400, 32768, 300
    tgrid = np.linspace(0.0, 1.5, m * M + 1)
    knots = tgrid[::M]
    y = np.array([0.9, 0.2, -0.4, 0.5])
    Sig, piH = 0.05, 1e3
    P = bh.LinPro([[-beta]], [0.0], [[math.sqrt(a)]])
    Pt = bh.LinPro([[-0.3]], [0.2], [[math.sqrt(a)]])               # a deliberately different auxiliary
    L, S = np.array([[1.0]]), np.array([[Sig]])
    HT, vT = bh.gpupdate(np.array([[piH]]), np.zeros(1), L, S, y[m:m + 1])
    H, v, segs = HT, vT, [None] * m
    for i in range(m - 1, -1, -1):
        segs[i] = bh.GuidedBridge(tgrid[i * M:(i + 1) * M + 1].copy(), P, Pt, v, H, ctx=ctx)
        H, v = bh.gpupdate(segs[i], L, S, y[i:i + 1])
    # exact law of (X(t_0), .., X(t_3))
    Lam, eta = np.zeros((m + 1, m + 1)), np.zeros(m + 1)
    for k in range(m + 1):
        Lam[k, k] += 1 / Sig; eta[k] += y[k] / Sig
    Lam[m, m] += 1 / piH
    for k in range(m):
        dl = knots[k + 1] - knots[k]
        phi, q = math.exp(-beta * dl), a * (1 - math.exp(-2 * beta * dl)) / (2 * beta)
        Lam[k, k] += phi * phi / q; Lam[k + 1, k + 1] += 1 / q
        Lam[k, k + 1] -= phi / q; Lam[k + 1, k] -= phi / q
    cov = np.linalg.inv(Lam)
    mean = cov @ eta
    sc = bh.SegChains(segs, v, np.sqrt(H), n, seed=23)

    def knot_values():
        cols = [sc.paths(i, 0, n)[0][:, 0, 0] for i in range(m)] + [sc.paths(m - 1, 0, n)[0][:, -1, 0]]
        return np.stack(cols, 1)
    X0 = knot_values()
    w_new = math.sqrt(0.5)
    sc.step(math.sqrt(1 - w_new ** 2), w_new, iters)
    X = knot_values()
    se = np.sqrt(np.diag(cov) / n)
    err = np.abs(X.mean(0) - mean)
    assert (err < 6 * se + 5e-3).all(), (X.mean(0), mean)
    # the Euler scheme inflates the variance at the right end of a guided segment by ~ a*dt/(2*Hd) (stiff guiding term
    # a/Hd ~ 15 near a noisy observation): 3.9 % at M = 100 steps per segment (measured 3-5 %), ~1 % here; MC error 0.8 %
    assert (np.abs(X.var(0) / np.diag(cov) - 1) < 0.035).all(), (X.var(0), np.diag(cov))
    emp = np.cov(X.T)
    assert np.abs(emp[0, 1] - cov[0, 1]) < 0.05 * math.sqrt(cov[0, 0] * cov[1, 1]) + 2e-3          # the knots are correlated as they should be
    # the first proposals (start at pi0's mean, wrong drift) are visibly off: the test has power
    assert np.abs(X0.mean(0) - mean).max() > 10 * err.max() or np.abs(X0.var(0) / np.diag(cov) - 1).max() > 0.3
    acc = sc.state()[1]
    assert 0.1 < acc.mean() / iters < 0.95


def test_device_built_guides_of_a_linear_target_sample_the_exact_smoothing_law(ctx):
    """The device guide kernel end to end, distribution level.  Same linear-Gaussian smoothing problem as
    above; the segments start with a deliberately wrong LinearAppr (B_i = -0.3, b_i = 0.06: the drift of a different linear
    process), so phase A needs the Metropolis-Hastings correction.  bhip_segchains_adapt_device then re-linearises every chain
    around its own running mean: bderiv of a LinPro is its B, so every chain's guide becomes the target's own
    (linearisation exact, log-likelihood ratio 0): phase B accepts every proposal and must STILL sample the closed-form law --
    which checks the chain-wise backward ODE, the gpupdate chain and pi0 = N(v, Hd) (mean and Cholesky factor) as built on
    the device, not just their agreement with the oracle."""
    beta, a = 0.8, 0.7
    m, M, n = 3, 400, 32768
    tgrid = np.linspace(0.0, 1.5, m * M + 1)
    knots = tgrid[::M]
    y = np.array([0.9, 0.2, -0.4, 0.5])
    Sig, piH = 0.05, 1e3
    P = bh.LinPro([[-beta]], [0.0], [[math.sqrt(a)]])
    L, S = np.array([[1.0]]), np.array([[Sig]])
    HT, vT = bh.gpupdate(np.array([[piH]]), np.zeros(1), L, S, y[m:m + 1])
    H, v, segs = HT, vT, [None] * m
    for i in range(m - 1, -1, -1):
        N1 = M + 1
        wrong = bh.LinearAppr(np.zeros((N1, 1)), np.full((N1, 1, 1), -0.3), np.full((N1, 1), 0.06), np.full((N1, 1, 1), math.sqrt(a)))
        segs[i] = bh.GuidedBridge(tgrid[i * M:(i + 1) * M + 1].copy(), P, wrong, v, H, ctx=ctx)
        H, v = bh.gpupdate(segs[i], L, S, y[i:i + 1])
    Lam, eta = np.zeros((m + 1, m + 1)), np.zeros(m + 1)
    for k in range(m + 1):
        Lam[k, k] += 1 / Sig; eta[k] += y[k] / Sig
    Lam[m, m] += 1 / piH
    for k in range(m):
        dl = knots[k + 1] - knots[k]
        phi, q = math.exp(-beta * dl), a * (1 - math.exp(-2 * beta * dl)) / (2 * beta)
        Lam[k, k] += phi * phi / q; Lam[k + 1, k + 1] += 1 / q
        Lam[k, k + 1] -= phi / q; Lam[k + 1, k] -= phi / q
    cov = np.linalg.inv(Lam)
    mean = cov @ eta
    se = np.sqrt(np.diag(cov) / n)
    sc = bh.SegChains(segs, v, np.sqrt(H), n, seed=29, mcnext=True)

    def knot_values():
        return np.stack([sc.paths(i, 0, n)[0][:, 0, 0] for i in range(m)] + [sc.paths(m - 1, 0, n)[0][:, -1, 0]], 1)

    def check(X):
        assert (np.abs(X.mean(0) - mean) < 6 * se + 5e-3).all(), (X.mean(0), mean)
        assert (np.abs(X.var(0) / np.diag(cov) - 1) < 0.035).all(), (X.var(0), np.diag(cov))
    w_new = math.sqrt(0.5)
    w_old = math.sqrt(1 - w_new ** 2)
    sc.step(w_old, w_new, 300)                       # phase A: wrong auxiliary, MH corrects
    accA = sc.state()[1].copy()
    assert 0.1 < accA.mean() / 300 < 0.95
    check(knot_values())
    sc.adapt_device(L, S, y[:m].reshape(m, 1), HT, vT, newblock=True, doaccept=True)
    g = sc.chain_guide(0, 123)
    assert np.array_equal(g["B"], np.full((M, 1, 1), -beta))                           # linearappr of a LinPro: B_i = B
    assert abs(g["mu"][0] - mean[0]) < 2e-3 and abs(g["chol"][0, 0] ** 2 / cov[0, 0] - 1) < 5e-3     # pi0 = the exact law of X(0)
    sc.step(w_old, w_new, 60)                        # phase B: exact guides
    ll, accB, _ = sc.state()
    assert np.array_equal(accB - accA, np.full(n, 60)) and np.abs(ll).max() < 1e-9
    check(knot_values())
