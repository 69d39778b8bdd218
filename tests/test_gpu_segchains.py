"""GPU tests of the joint Metropolis-Hastings over chained segments (bhip_segchains_*), the loop of
supplements/smoothing/smoothing.jl:99-213 / test/smoothing.jl:73-92, against its oracle twin bo_smooth_mcmc.

Bit-exact (==) for polynomial drifts: current paths of every segment, Wiener paths, per-segment log-likelihoods, the
pCN-moved starting point, acceptance counts and the per-chain mcnext! state (mean, m2) after every iteration count.
"""
import numpy as np
import pytest

import bridgehip as bh
import oracle as o

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    return bh.default_context(0)


def build_segments(ctx, kind, m=3, M=40):
    """m chained GuidedBridge segments, backward recursion of (Hdiamond, v) through gpupdate (test/smoothing.jl:73-85)"""
    rng = np.random.default_rng(5)
    if kind == "linpro2":         # d = m' = 2: line layout, wave-specialised kernel
        d = 2
        B = np.array([[-1, 0.1], [-0.2, -1]])
        sig = 2 * np.array([[-0.212887, 0.0687025], [0.193157, 0.388997]])
        P, Pt = bh.LinPro(B, [0.02, 0.03], sig), bh.LinPro(0.8 * B, [0.0, 0.0], sig)
        model, par, aux, apar = o.MODEL_LINPRO, o.linpro_par(B, [0.02, 0.03], sig), o.AUX_LINPRO, o.linpro_par(0.8 * B, [0.0, 0.0], sig)
        L, Sig = np.array([[1.0, 0.0]]), np.array([[0.05]])
    elif kind == "ou1":           # d = m' = 1
        d = 1
        P, Pt = bh.LinPro([[-0.8]], [0.0], [[0.8]]), bh.LinPro([[-0.8]], [0.2], [[0.8]])
        model, par, aux, apar = o.MODEL_LINPRO, o.linpro_par([[-0.8]], [0.0], [[0.8]]), o.AUX_LINPRO, o.linpro_par([[-0.8]], [0.2], [[0.8]])
        L, Sig = np.array([[1.0]]), np.array([[0.1]])
    else:                         # Lorenz, d = m' = 3: slot layout, path-per-lane kernel   (test/smoothing.jl:19-21)
        d = 3
        P = bh.Lorenz((10.0, 20.0, 8 / 3), (3.0, 3.0, 3.0))
        Pt = bh.LinPro(-np.eye(3), [0.0, 0.0, 0.0], 3.0 * np.eye(3))
        model, par = o.MODEL_LORENZ, [10.0, 20.0, 8 / 3, 3.0, 3.0, 3.0]
        aux, apar = o.AUX_LINPRO, o.linpro_par(-np.eye(3), [0.0, 0.0, 0.0], 3.0 * np.eye(3))
        L, Sig = np.eye(3), np.eye(3)
    mo = L.shape[0]
    tgrid = np.linspace(0, 0.3 * m, m * M + 1)
    obs = rng.standard_normal((m + 1, mo))
    H, v = bh.gpupdate(np.diag([np.inf] * d), np.zeros(d), np.eye(d), 0.5 * np.eye(d), np.concatenate([obs[m], np.zeros(d)])[:d])
    segs, refs = [None] * m, [None] * m
    for i in range(m - 1, -1, -1):
        tt = tgrid[i * M:(i + 1) * M + 1].copy()
        segs[i] = bh.GuidedBridge(tt, P, Pt, v, H, ctx=ctx)
        refs[i] = o.proposal_hv(tt, d, d, model, par, aux, apar, segs[i].Hd, segs[i].V)
        H, v = bh.gpupdate(segs[i], L, Sig, obs[i])
    # pi0 = Gaussian(v, Hdiamond) after the last update (smoothing.jl:97-99)
    return segs, refs, v, np.linalg.cholesky((H + H.T) / 2), d


@pytest.mark.parametrize("kind", ["linpro2", "ou1", "lorenz"])
def test_joint_mh_over_segments_equals_oracle(ctx, kind):
    segs, refs, mu, chol, d = build_segments(ctx, kind)
    n, iters = 200, 7
    rng = np.random.default_rng(0)
    rho_ = np.exp(-0.5 * rng.exponential(size=iters))          # smoothing.jl:171  rho_ = exp(-alpha*randexp())
    w_new, w_old = np.sqrt(rho_), np.sqrt(1 - rho_)
    sc = bh.SegChains(segs, mu, chol, n, seed=17, path0=40, mcnext=True)
    sc.step(w_old[:3], w_new[:3])
    sc.step(w_old[3:], w_new[3:])                              # iteration counters continue across calls
    ll, acc, y0 = sc.state()
    total_acc = 0
    for p in (0, 63, 64, 199):
        r = o.smooth_mcmc(refs, mu, chol, w_old, w_new, 17, 40 + p, stats=True)
        for i in range(len(segs)):
            X, W = sc.paths(i, p, 1)
            assert np.array_equal(X[0], r["X"][i]), (kind, p, i, np.abs(X[0] - r["X"][i]).max())
            assert np.array_equal(W[0], r["W"][i])
            mean, m2, cnt = sc.mcstats(i, p)
            assert cnt == iters == r["n"]
            assert np.array_equal(mean, r["mean"][i]) and np.array_equal(m2, r["m2"][i])
        assert np.array_equal(ll[:, p], r["ll"]) and acc[p] == r["acc"] and np.array_equal(y0[p], r["y0"])
        total_acc += r["acc"]
    assert 0 < acc.sum() < n * iters
    # joints: every current path is continuous across the segment boundaries and starts at the chain's y0
    X0, _ = sc.paths(0, 0, n)
    assert np.array_equal(X0[:, 0, :], y0)
    for i in range(len(segs) - 1):
        Xa, _ = sc.paths(i, 0, n)
        Xb, _ = sc.paths(i + 1, 0, n)
        assert np.array_equal(Xa[:, -1, :], Xb[:, 0, :])


@pytest.mark.parametrize("kind,pooled", [("lorenz", True), ("linpro2", True), ("lorenz", False), ("linpro2", False), ("ou1", False)])
def test_commit_stream_overlap_changes_nothing(ctx, kind, pooled):
    """The commit + mcnext! of iteration t runs on the library's second stream beside the proposals of iteration t + 1
    (bhip_segchains.inc): nine iterations in ONE call -- commits overlapped, buffer reuse waited for -- leave exactly the state
    that nine single-iteration calls (each joined before it returns) leave, at a size where the kernels really run side by
    side.  pooled: plain SoA paths, two proposal buffers, commit copy, the pooled kernels behind the commit on the second
    stream; otherwise (d <= 3): time-blocked paths in parity halves, mcnext! alone on the second stream."""
    segs, refs, mu, chol, d = build_segments(ctx, kind, m=3, M=64)
    n, iters = 20000, 9
    rng = np.random.default_rng(3)
    rho_ = np.exp(-0.5 * rng.exponential(size=iters))
    w_new, w_old = np.sqrt(rho_), np.sqrt(1 - rho_)
    a = bh.SegChains(segs, mu, chol, n, seed=23, mcnext=True, pooled=pooled)
    b = bh.SegChains(segs, mu, chol, n, seed=23, mcnext=True, pooled=pooled)
    a.step(w_old, w_new)
    for it in range(iters):
        b.step(w_old[it:it + 1], w_new[it:it + 1])
    for x, y in zip(a.state(), b.state()):
        assert np.array_equal(x, y)
    assert 0 < a.state()[1].sum() < n * iters
    for i in range(len(segs)):
        Xa, Wa = a.paths(i, 0, n)
        Xb, Wb = b.paths(i, 0, n)
        assert np.array_equal(Xa, Xb) and np.array_equal(Wa, Wb)
        for p in (0, 777, n - 1):
            for u, v in zip(a.mcstats(i, p), b.mcstats(i, p)):
                assert np.array_equal(u, v)
        if pooled:
            for u, v in zip(a.pooled_stats(i), b.pooled_stats(i)):
                assert np.array_equal(u, v)
    # and against the oracle for one chain
    r = o.smooth_mcmc(refs, mu, chol, w_old, w_new, 23, 777, stats=True)
    for i in range(len(segs)):
        assert np.array_equal(a.paths(i, 777, 1)[0][0], r["X"][i])
        mean, m2, cnt = a.mcstats(i, 777)
        assert cnt == iters and np.array_equal(mean, r["mean"][i]) and np.array_equal(m2, r["m2"][i])


@pytest.mark.parametrize("M,n", [(1, 1), (3, 65), (14, 64), (15, 130), (16, 63), (17, 129), (31, 5), (32, 200)])
def test_time_blocked_paths_at_block_edges(ctx, M, n):
    """Grids of N = M + 1 points around the sixteen-point blocks of the time-blocked paths (a block that ends exactly at the end
    point, one that holds the end point alone, a grid shorter than a block) and chain counts around the 64-chain groups: every
    chain equals its oracle twin -- paths, W, statistics, decisions."""
    segs, refs, mu, chol, d = build_segments(ctx, "lorenz", m=2, M=M)
    iters = 5
    rng = np.random.default_rng(M)
    rho_ = np.exp(-0.5 * rng.exponential(size=iters))
    w_new, w_old = np.sqrt(rho_), np.sqrt(1 - rho_)
    sc = bh.SegChains(segs, mu, chol, n, seed=9, path0=3, mcnext=True)
    sc.step(w_old, w_new)
    ll, acc, y0 = sc.state()
    for p in sorted({0, n // 2, n - 1}):
        r = o.smooth_mcmc(refs, mu, chol, w_old, w_new, 9, 3 + p, stats=True)
        assert acc[p] == r["acc"] and np.array_equal(ll[:, p], r["ll"]) and np.array_equal(y0[p], r["y0"])
        for i in range(2):
            X, W = sc.paths(i, p, 1)
            assert np.array_equal(X[0], r["X"][i]) and np.array_equal(W[0], r["W"][i])
            mean, m2, cnt = sc.mcstats(i, p)
            assert cnt == iters and np.array_equal(mean, r["mean"][i]) and np.array_equal(m2, r["m2"][i])


def test_time_blocked_paths_with_user_process_noise_spec_2_and_fused_build(ctx):
    """The other routes into the wave-specialised kernel take the time-blocked path stores too: a hipRTC process (the same kernel
    template, compiled at run time, its LDS sized by the launch code of that route) reproduces the built-in Lorenz bit for bit;
    the full-resolution noise stream (BHIP_OPT_NOISE_SPEC = 2) equals its oracle twin; the fused-arithmetic build agrees to its
    stated tolerance on the first proposal."""
    import torch
    m, M, n, iters = 2, 37, 150, 4
    th, sg = (10.0, 20.0, 8 / 3), (3.0, 3.0, 3.0)
    rng = np.random.default_rng(2)
    tgrid = np.linspace(0, 0.05 * m, m * M + 1)
    obs = np.array([1.5, -1.5, 25.0]) + rng.standard_normal((m + 1, 3))
    rho_ = np.exp(-0.5 * rng.exponential(size=iters))
    w_new, w_old = np.sqrt(rho_), np.sqrt(1 - rho_)

    def build(c, P):
        Pt = bh.LinPro(-np.eye(3), np.zeros(3), np.diag(sg))
        H, v = bh.gpupdate(np.diag([np.inf] * 3), np.zeros(3), np.eye(3), 0.5 * np.eye(3), obs[m])
        segs, refs = [None] * m, [None] * m
        for i in range(m - 1, -1, -1):
            tt = tgrid[i * M:(i + 1) * M + 1].copy()
            segs[i] = bh.GuidedBridge(tt, P, Pt, v, H, ctx=c)
            refs[i] = o.proposal_hv(tt, 3, 3, o.MODEL_LORENZ, [*th, *sg], o.AUX_LINPRO, o.linpro_par(-np.eye(3), np.zeros(3), np.diag(sg)), segs[i].Hd, segs[i].V)
            H, v = bh.gpupdate(segs[i], np.eye(3), np.eye(3), obs[i])
        return segs, refs, v, np.linalg.cholesky((H + H.T) / 2)

    def run(c, P, its=iters):
        segs, refs, mu, chol = build(c, P)
        sc = bh.SegChains(segs, mu, chol, n, seed=4, path0=1, mcnext=True)
        sc.step(w_old[:its], w_new[:its])
        return sc, refs, mu, chol

    src = "o[0] = par[0]*(x[1] - x[0]); o[1] = x[0]*(par[1] - x[2]) - x[1]; o[2] = x[0]*x[1] - par[2]*x[2];"
    a, refs, mu, chol = run(ctx, bh.Lorenz(th, sg))
    b, _, _, _ = run(ctx, bh.UserProcess(3, src, list(th), np.diag(sg), ctx=ctx))
    for x, y in zip(a.state(), b.state()):
        assert np.array_equal(x, y)
    for i in range(m):
        assert np.array_equal(a.paths(i, 0, n)[0], b.paths(i, 0, n)[0])
        for u, v in zip(a.mcstats(i, n - 1), b.mcstats(i, n - 1)):
            assert np.array_equal(u, v)
    r = o.smooth_mcmc(refs, mu, chol, w_old, w_new, 4, 1 + 77, stats=True)
    assert np.array_equal(a.paths(1, 77, 1)[0][0], r["X"][1]) and a.state()[1][77] == r["acc"]

    c2 = bh.Context(0)
    c2.set_option(bh.OPT_NOISE_SPEC, 2)
    v2, refs2, mu2, chol2 = run(c2, bh.Lorenz(th, sg))
    with o.noise_spec(2):
        for p in (0, 64, n - 1):
            r = o.smooth_mcmc(refs2, mu2, chol2, w_old, w_new, 4, 1 + p, stats=True)
            for i in range(m):
                X, W = v2.paths(i, p, 1)
                assert np.array_equal(X[0], r["X"][i]) and np.array_equal(W[0], r["W"][i])
                mean, m2, cnt = v2.mcstats(i, p)
                assert np.array_equal(mean, r["mean"][i]) and np.array_equal(m2, r["m2"][i])
            assert v2.state()[1][p] == r["acc"]
    assert not np.array_equal(v2.paths(0, 0, 1)[1], a.paths(0, 0, 1)[1])

    cf = bh.Context(0)
    cf.set_option(bh.OPT_FUSED_ARITHMETIC, 1)
    f1, _, _, _ = run(cf, bh.Lorenz(th, sg), its=1)
    e1, _, _, _ = run(ctx, bh.Lorenz(th, sg), its=1)
    same = f1.state()[1] == e1.state()[1]                    # chains that took the same first decision hold the same proposal (or none)
    assert same.mean() > 0.95
    for i in range(m):
        Xf, Wf = f1.paths(i, 0, n)
        Xe, We = e1.paths(i, 0, n)
        assert np.abs(Wf[same] - We[same]).max() <= 1e-13                      # (the pCN mix rho*W + sqrt(1 - rho^2)*W2 is contracted too)
        assert np.abs(Xf[same] - Xe[same]).max() <= 1e-9 * (1 + np.abs(Xe).max())
    torch.cuda.synchronize()


@pytest.mark.parametrize("defer", ["1 1", "2 1", "3 2", "4 4", "8 4", "5 5", "8 8", "12 4", "15 1"])
def test_deferred_statistics_equal_a_pass_per_iteration(ctx, defer, monkeypatch):
    """mcnext! of the time-blocked paths is applied every K iterations to the K current paths a ring of K + L buffers has kept, the
    pass running beside the first L iterations of the next batch (bhip_segchains.inc): whatever K and L (BHIP_SEG_DEFER at creation),
    whatever the split of the iterations into calls (each call ends with a pass over what is pending), the chains and their
    statistics are those of the oracle's loop, which updates every iteration -- bit for bit."""
    segs, refs, mu, chol, d = build_segments(ctx, "lorenz", m=2, M=40)
    n, iters = 9000, 23
    rng = np.random.default_rng(8)
    rho_ = np.exp(-0.5 * rng.exponential(size=iters))
    w_new, w_old = np.sqrt(rho_), np.sqrt(1 - rho_)
    monkeypatch.setenv("BHIP_SEG_DEFER", defer)
    sc = bh.SegChains(segs, mu, chol, n, seed=31, path0=2, mcnext=True)
    monkeypatch.delenv("BHIP_SEG_DEFER")
    cuts = [0, 13, 14, 17, iters]                                  # 13 iterations in one call, then 1, 3 and 6
    for a, b in zip(cuts[:-1], cuts[1:]):
        sc.step(w_old[a:b], w_new[a:b])
        assert sc.mcstats(0, 0)[2] == b
    ll, acc, y0 = sc.state()
    assert 0 < acc.sum() < n * iters
    for p in (0, 4444, n - 1):
        r = o.smooth_mcmc(refs, mu, chol, w_old, w_new, 31, 2 + p, stats=True)
        assert acc[p] == r["acc"] and np.array_equal(ll[:, p], r["ll"])
        for i in range(2):
            assert np.array_equal(sc.paths(i, p, 1)[0][0], r["X"][i])
            mean, m2, cnt = sc.mcstats(i, p)
            assert cnt == iters and np.array_equal(mean, r["mean"][i]) and np.array_equal(m2, r["m2"][i])


@pytest.mark.parametrize("name", ["fhn_partialbridge_extreme", "fhn2_nuh_full", "nclar_firstcomponent", "ouproc_nuh", "linpro2_guidedbridge"])
def test_joint_mh_over_partial_bridge_segments(ctx, name):
    """The loop takes any guided proposals: PartialBridge (L, M, mu), PartialBridgeNuH, a sin-drift process (NCLAR, d = 3 with scalar
    noise) -- every guide form of the wave-specialised kernel writes the time-blocked paths and hands the end point on.  Two segments on
    the case's own grid (the second starts where the first ends), pi0 around the case's starting point, deferred statistics: == the oracle."""
    import problems
    c = [c for c in problems.cases(77) if c.name == name][0]
    segs = [c.bh_proposal(bh, ctx), c.bh_proposal(bh, ctx)]
    ref = c.oracle_proposal()
    refs = [ref, ref]
    mu, chol = c.x0, 0.05 * np.eye(c.d)
    n, iters = 130, 6
    rng = np.random.default_rng(1)
    rho_ = np.exp(-0.5 * rng.exponential(size=iters))
    w_new, w_old = np.sqrt(rho_), np.sqrt(1 - rho_)
    sc = bh.SegChains(segs, mu, chol, n, seed=12, path0=9, mcnext=True)
    sc.step(w_old[:5], w_new[:5])
    sc.step(w_old[5:], w_new[5:])
    ll, acc, y0 = sc.state()
    for p in (0, 64, n - 1):
        r = o.smooth_mcmc(refs, mu, chol, w_old, w_new, 12, 9 + p, stats=True)
        assert acc[p] == r["acc"] and np.array_equal(y0[p], r["y0"]), (name, p)
        for i in range(2):
            X, W = sc.paths(i, p, 1)
            assert np.array_equal(W[0], r["W"][i]), (name, p, i)
            assert np.array_equal(X[0], r["X"][i]), (name, p, i, np.abs(X[0] - r["X"][i]).max())
            mean, m2, cnt = sc.mcstats(i, p)
            assert cnt == iters and np.array_equal(mean, r["mean"][i]) and np.array_equal(m2, r["m2"][i])
        assert np.array_equal(ll[:, p], r["ll"])


@pytest.mark.parametrize("defer", [None, "3 1", "7 3", "4 4", "8 8"])
def test_deferred_statistics_over_many_iterations(ctx, defer, monkeypatch):
    """600 iterations in calls of irregular length: the ring of path buffers is walked hundreds of times in every pattern of accepts
    and rejects; chains, statistics and acceptance counts equal the oracle's."""
    segs, refs, mu, chol, d = build_segments(ctx, "lorenz", m=2, M=20)
    n, iters = 200, 600
    rng = np.random.default_rng(77)
    rho_ = np.exp(-0.5 * rng.exponential(size=iters))
    w_new, w_old = np.sqrt(rho_), np.sqrt(1 - rho_)
    if defer:
        monkeypatch.setenv("BHIP_SEG_DEFER", defer)
    sc = bh.SegChains(segs, mu, chol, n, seed=5, path0=0, mcnext=True)
    if defer:
        monkeypatch.delenv("BHIP_SEG_DEFER")
        assert sc.statistics_info() == (int(defer.split()[0]), sum(int(x) for x in defer.split()))
    else:
        assert sc.statistics_info() == (12, 16)    # the default ring since round 6 (8 + 8 at the end of round 5, 4 + 4 before)
        every = bh.SegChains(segs, mu, chol, n, seed=5, path0=0, mcnext=True, stats_every_iteration=True)   # BHIP_SEGCHAINS_STATS_EVERY_ITERATION
        assert every.statistics_info() == (1, 2)
        every.step(w_old[:50], w_new[:50])
    a = 0
    while a < iters:
        b = min(iters, a + int(rng.integers(1, 40)))
        sc.step(w_old[a:b], w_new[a:b])
        a = b
    ll, acc, y0 = sc.state()
    for p in (0, 101, n - 1):
        r = o.smooth_mcmc(refs, mu, chol, w_old, w_new, 5, p, stats=True)
        assert acc[p] == r["acc"] and 0 < r["acc"] < iters
        for i in range(2):
            assert np.array_equal(sc.paths(i, p, 1)[0][0], r["X"][i])
            mean, m2, cnt = sc.mcstats(i, p)
            assert cnt == iters and np.array_equal(mean, r["mean"][i]) and np.array_equal(m2, r["m2"][i])
    if not defer:
        r50 = o.smooth_mcmc(refs, mu, chol, w_old[:50], w_new[:50], 5, 101, stats=True)
        for i in range(2):
            assert np.array_equal(every.paths(i, 101, 1)[0][0], r50["X"][i])
            for u, v in zip(every.mcstats(i, 101)[:2], (r50["mean"][i], r50["m2"][i])):
                assert np.array_equal(u, v)


@pytest.mark.parametrize("kind", ["lorenz", "linpro2", "ou1"])
def test_time_blocked_paths_equal_plain_paths(ctx, kind, monkeypatch):
    """d <= 3 without pooled statistics keeps the segments' paths time-blocked in parity halves (accept = parity flip, mcnext!
    reads the current halves); BHIP_SEG_PLAIN_X=1 at creation keeps the plain SoA paths with the commit copy.  Same chains,
    bit for bit, also with a grid whose length is not a multiple of the block (N = 51) and a chain count that is no multiple
    of 64; the means-only statistics and no statistics at all take the same route."""
    segs, refs, mu, chol, d = build_segments(ctx, kind, m=3, M=50)
    n, iters = 333, 8
    rng = np.random.default_rng(11)
    rho_ = np.exp(-0.5 * rng.exponential(size=iters))
    w_new, w_old = np.sqrt(rho_), np.sqrt(1 - rho_)
    for kw in ({"mcnext": True}, {"mcnext_mean_only": True}, {}):
        tb = bh.SegChains(segs, mu, chol, n, seed=3, path0=7, **kw)
        monkeypatch.setenv("BHIP_SEG_PLAIN_X", "1")
        pl = bh.SegChains(segs, mu, chol, n, seed=3, path0=7, **kw)
        monkeypatch.delenv("BHIP_SEG_PLAIN_X")
        for c in (tb, pl):
            c.step(w_old[:5], w_new[:5])
        X_mid = [tb.paths(i, 0, n)[0] for i in range(3)]          # (materialises the plain-layout current paths in between)
        for i in range(3):
            assert np.array_equal(X_mid[i], pl.paths(i, 0, n)[0])
        for c in (tb, pl):
            c.step(w_old[5:], w_new[5:])
        for x, y in zip(tb.state(), pl.state()):
            assert np.array_equal(x, y)
        for i in range(3):
            Xa, Wa = tb.paths(i, 0, n)
            Xb, Wb = pl.paths(i, 0, n)
            assert np.array_equal(Xa, Xb) and np.array_equal(Wa, Wb)
            assert not np.array_equal(Xa, X_mid[i])
            if kw:
                for p in (0, 64, n - 1):
                    for u, v in zip(tb.mcstats(i, p), pl.mcstats(i, p)):
                        assert (u is None and v is None) or np.array_equal(u, v)


def test_segchains_argument_checks(ctx):
    segs, refs, mu, chol, d = build_segments(ctx, "ou1", m=2)
    other = bh.GuidedBridge(np.linspace(0, 1, 11), bh.LinPro([[-0.8]], [0.0], [[0.8]]), bh.LinPro([[-0.8]], [0.2], [[0.8]]), [0.1], ctx=ctx)
    with pytest.raises(bh.BridgeError, match="number of grid points"):
        bh.SegChains([segs[0], other], mu, chol, 8)
    sc = bh.SegChains(segs, mu, chol, 8)              # without mcnext
    sc.step(0.9, np.sqrt(1 - 0.81), 2)
    with pytest.raises(bh.BridgeError, match="MCNEXT"):
        sc.mcstats(0, 0)
    # the pCN weights must lie on the unit circle (advisor r2: anything else silently targets another law)
    for wo, wn in ((0.9, 0.5), (float("nan"), 0.1), (1.0, 1e-3)):
        with pytest.raises(bh.BridgeError, match="w_old\\^2 \\+ w_new\\^2 = 1"):
            sc.step(wo, wn, 1)
    sc.step([0.9, 0.5], [np.sqrt(1 - 0.81), np.sqrt(0.75)])          # per-iteration weights, each pair on the circle


def test_segchains_create_destroy_does_not_leak(ctx):
    """advisor r2: segment 0's parity / count arrays were never released (5 * ld bytes per ensemble)"""
    import torch
    segs, refs, mu, chol, d = build_segments(ctx, "ou1", m=2)
    n = 1 << 20                                           # 5 MiB of cur + acc per ensemble
    def cycle():
        sc = bh.SegChains(segs, mu, chol, n)
        del sc
    cycle(); torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(40):
        cycle()
    torch.cuda.synchronize()
    assert free0 - torch.cuda.mem_get_info()[0] < 64 << 20   # 40 leaked ensembles would be 200 MiB


def test_pooled_statistics_over_chains_and_iterations(ctx):
    """BHIP_SEGCHAINS_POOLED: the device-resident mcnext over chains x iterations equals mean / scatter of all current paths"""
    segs, refs, mu, chol, d = build_segments(ctx, "linpro2", m=2)
    n, iters = 300, 5
    sc = bh.SegChains(segs, mu, chol, n, seed=5, pooled=True)
    samples = [[], []]
    for it in range(iters):
        sc.step(0.8, 0.6, 1)
        for i in range(2):
            samples[i].append(sc.paths(i, 0, n)[0])
    for i in range(2):
        S = np.concatenate(samples[i])                       # [iters*n, N, d]
        mean, m2, cnt = sc.pooled_stats(i)
        assert cnt == n * iters
        assert np.allclose(mean, S.mean(0), rtol=1e-12, atol=1e-13)
        dev = S - S.mean(0)
        assert np.allclose(m2, np.einsum("sni,snj->nij", dev, dev), rtol=1e-10, atol=1e-11)


@pytest.mark.parametrize("d", [16, 5])
def test_joint_mh_over_segments_at_large_dimension(ctx, d):
    """round 3: multi-segment chains on the MFMA tile kernel (d = 16; d = 5 zero padded): per-chain starts from the previous
    segment's end point, one noise stream per segment, the accept deferred to the joint decision, the pCN move of a d-vector start,
    running means and the full per-chain mcnext! state.  Against bo_smooth_mcmc at the tile kernel's tolerance (pre-inverted guide matrix, fused MFMA accumulation):
    identical accept decisions, Wiener paths bit for bit, paths 1e-9, ll 1e-8."""
    rng = np.random.default_rng(11)
    m, M = 3, 30
    G = rng.standard_normal((d, d)) / np.sqrt(d); G2 = rng.standard_normal((d, d)) / np.sqrt(d)
    B, sig = -np.eye(d) + 0.1 * G, 0.5 * np.eye(d) + 0.05 * G2
    P, Pt = bh.LinPro(B, np.zeros(d), sig), bh.LinPro(-np.eye(d), np.zeros(d), sig)
    par, apar = o.linpro_par(B, np.zeros(d), sig), o.linpro_par(-np.eye(d), np.zeros(d), sig)
    L, Sig = np.eye(d), 0.3 * np.eye(d)
    tgrid = np.linspace(0, 0.2 * m, m * M + 1)
    obs = 0.3 * rng.standard_normal((m + 1, d))
    H, v = bh.gpupdate(np.diag([np.inf] * d), np.zeros(d), np.eye(d), 0.5 * np.eye(d), obs[m])
    segs, refs = [None] * m, [None] * m
    for i in range(m - 1, -1, -1):
        tt = tgrid[i * M:(i + 1) * M + 1].copy()
        segs[i] = bh.GuidedBridge(tt, P, Pt, v, H, ctx=ctx)
        refs[i] = o.proposal_hv(tt, d, d, o.MODEL_LINPRO, par, o.AUX_LINPRO, apar, segs[i].Hd, segs[i].V)
        H, v = bh.gpupdate(segs[i], L, Sig, obs[i])
    chol = np.linalg.cholesky((H + H.T) / 2)
    n, iters = 100, 6
    w_new = np.sqrt(rng.uniform(0.05, 0.4, iters)); w_old = np.sqrt(1 - w_new ** 2)
    sc = bh.SegChains(segs, v, chol, n, seed=23, path0=5, mcnext_mean_only=True)
    full = bh.SegChains(segs, v, chol, n, seed=23, path0=5, mcnext=True)     # mcnext! literally: d + d*d doubles per chain and grid point
    for c in (sc, full):
        c.step(w_old[:2], w_new[:2])
        c.step(w_old[2:], w_new[2:])
    ll, acc, y0 = sc.state()
    llf, accf, y0f = full.state()
    assert np.array_equal(ll, llf) and np.array_equal(acc, accf) and np.array_equal(y0, y0f)
    for p in (0, 15, 16, 99):
        r = o.smooth_mcmc(refs, v, chol, w_old, w_new, 23, 5 + p, stats=True)
        assert acc[p] == r["acc"], (p, acc[p], r["acc"])
        assert np.abs(y0[p] - r["y0"]).max() <= 1e-12 * (1 + np.abs(r["y0"]).max())
        for i in range(m):
            X, W = sc.paths(i, p, 1)
            assert np.array_equal(W[0], r["W"][i]) or np.abs(W[0] - r["W"][i]).max() <= 1e-14 * (1 + np.abs(r["W"][i]).max())
            assert np.abs(X[0] - r["X"][i]).max() <= 1e-9 * (1 + np.abs(r["X"][i]).max()), (p, i)
            assert abs(ll[i, p] - r["ll"][i]) <= 1e-8 * (1 + abs(r["ll"][i]))
            mean, _, cnt = sc.mcstats(i, p)
            assert cnt == iters and np.abs(mean - r["mean"][i]).max() <= 1e-9 * (1 + np.abs(r["mean"][i]).max())
            meanf, m2f, cntf = full.mcstats(i, p)
            assert cntf == iters and np.array_equal(meanf, mean)
            assert np.abs(m2f - r["m2"][i]).max() <= 1e-9 * (1 + np.abs(r["m2"][i]).max())
    assert 0 < acc.sum() < n * iters
    for i in range(m - 1):                                                # continuity at the joints
        Xa, _ = sc.paths(i, 0, n)
        Xb, _ = sc.paths(i + 1, 0, n)
        assert np.array_equal(Xa[:, -1, :], Xb[:, 0, :])


@pytest.mark.parametrize("d", [5, 16])
def test_adaptive_loop_with_shared_guides_above_three_dimensions(ctx, d):
    """The adaptation block of supplements/smoothing/smoothing.jl:130-160 at d > 3 (the shared-guide half): LinearAppr
    auxiliaries by grid index over a LinPro target (src/linpro.jl:181-204) on the path-per-lane kernels (d = 5) and the tile kernel
    (d = 16); every `adaptit` iterations the segments are re-linearised around running means (SegChains.adapt: host-built guides handed
    over with bhip_segchains_set_proposals), pi0 replaced, newblock / doaccept as in the script.  For a linear target the linearisation
    does not depend on the point, so the ensemble's shared guides are every chain's own and the run is bo_smooth_adaptive's chain by
    chain: Wiener paths bit for bit, paths 1e-9, starts, running means, counts; the guides the adaptation built == the oracle's.
    (Per-chain device-built guides stay at d <= 3: bhip_segchains_adapt_device.)"""
    rng = np.random.default_rng(3)
    m, M, n = 3, 24, 70
    G = rng.standard_normal((d, d)) / np.sqrt(d)
    B, sig, mu_t = -0.8 * np.eye(d) + 0.3 * G, 0.5 * np.eye(d), 0.2 * rng.standard_normal(d)
    P = bh.LinPro(B, mu_t, sig)
    par = o.linpro_par(B, mu_t, sig)
    mo = 2
    L = np.zeros((mo, d)); L[0, 0] = L[1, 2] = 1.0                     # two components observed
    Sig = 0.2 * np.eye(mo)
    tgrid = np.linspace(0.0, 0.15 * m, m * M + 1)
    obs = 0.5 * rng.standard_normal((m + 1, mo))
    HT, vT = bh.gpupdate(1e3 * np.eye(d), np.zeros(d), L, Sig, obs[m])
    tts = np.stack([tgrid[i * M:(i + 1) * M + 1] for i in range(m)])
    Y0 = 0.3 * np.sin(np.arange(m * (M + 1) * d).reshape(m, M + 1, d) * 0.37)      # any first linearisation paths
    H, v, segs = HT, vT, [None] * m
    for i in range(m - 1, -1, -1):
        segs[i] = bh.GuidedBridge(tts[i].copy(), P, bh.linearappr(Y0[i]), v, H, ctx=ctx)
        H, v = bh.gpupdate(segs[i], L, Sig, obs[i])
    chol = o.chol_lower(H)
    adaptit, iters = 4, 10
    w_new = np.sqrt(np.full(iters, 0.2)); w_old = np.sqrt(1 - w_new ** 2)
    sc = bh.SegChains(segs, v, chol, n, seed=31, mcnext_mean_only=True)
    sc.step(w_old[:adaptit - 1], w_new[:adaptit - 1])
    means = [sc.mcstats(i, 11)[0] for i in range(m)]                    # (any chain's means: the linearisation of a linear target is the target)
    mu1, H1 = sc.adapt(P, L, Sig, obs[:m], HT, vT, means=means, newblock=True, doaccept=True)
    sc.step(w_old[adaptit - 1:2 * adaptit - 1], w_new[adaptit - 1:2 * adaptit - 1])
    means = [sc.mcstats(i, 11)[0] for i in range(m)]
    sc.adapt(P, L, Sig, obs[:m], HT, vT, means=means, newblock=True, doaccept=False)
    sc.step(w_old[2 * adaptit - 1:], w_new[2 * adaptit - 1:])
    ll, acc, y0 = sc.state()
    r1 = o.smooth_adaptive(o.MODEL_LINPRO, d, d, par, tts, Y0, L, Sig, obs[:m], HT, vT, w_old[:adaptit], w_new[:adaptit], adaptit, 10 ** 6, 31, 11)
    assert np.abs(mu1 - r1["mu"]).max() <= 1e-11 * (1 + np.abs(r1["mu"]).max()) and np.abs(H1 - r1["H"]).max() <= 1e-11 * (1 + np.abs(r1["H"]).max())
    for i in range(m):
        assert np.abs(sc.pos[i].Hd - r1["Hd"][i]).max() <= 1e-11 * (1 + np.abs(r1["Hd"][i]).max())
        assert np.abs(sc.pos[i].V - r1["V"][i]).max() <= 1e-11 * (1 + np.abs(r1["V"][i]).max())
    for p in (0, 11, 63, 64, n - 1):
        r = o.smooth_adaptive(o.MODEL_LINPRO, d, d, par, tts, Y0, L, Sig, obs[:m], HT, vT, w_old, w_new, adaptit, 10 ** 6, 31, p)
        assert acc[p] == r["acc"], (p, acc[p], r["acc"])
        assert np.abs(y0[p] - r["y0"]).max() <= 1e-10 * (1 + np.abs(r["y0"]).max())
        for i in range(m):
            X, W = sc.paths(i, p, 1)
            assert np.array_equal(W[0], r["W"][i]) or np.abs(W[0] - r["W"][i]).max() <= 1e-14 * (1 + np.abs(r["W"][i]).max())
            assert np.abs(X[0] - r["X"][i]).max() <= 1e-9 * (1 + np.abs(r["X"][i]).max()), (p, i)
            assert abs(ll[i, p] - r["ll"][i]) <= 1e-8 * (1 + abs(r["ll"][i]))
            mean, _, cnt = sc.mcstats(i, p)
            assert cnt == iters and np.abs(mean - r["mean"][i]).max() <= 1e-9 * (1 + np.abs(r["mean"][i]).max())
    for i in range(m - 1):                                                # continuity at the joints
        Xa, _ = sc.paths(i, 0, n)
        Xb, _ = sc.paths(i + 1, 0, n)
        assert np.array_equal(Xa[:, -1, :], Xb[:, 0, :])


def test_large_segments_are_placed_and_results_do_not_depend_on_it(ctx):
    """Segments of 1 GiB or more keep W and Xo in two contiguous allocations (bhip_api.hip chains_alloc_state); bhip_segchains_init
    places every such pair (different 96-GiB pieces of the device memory, measured: chains_place) before the ensemble's state is set
    up.  Decisions, log-likelihoods, starts and paths are those of the ensemble that was not placed (BHIP_OPT_TUNE_PLACEMENT = 0).
    (Pooled statistics: the ensembles whose proposals go to the plain Xo buffers -- with time-blocked paths the hot loop does not touch
    Xo and nothing is placed.)"""
    import ctypes as C
    import torch
    segs, refs, mu, chol, d = build_segments(ctx, "linpro2", m=2, M=512)     # 2 x 513 grid points, d = m' = 2
    n = 72000                                                                # x (2 x 33 lines x 128 B + 513 x 2 x 8 B) = 1.2 GB per segment
    w_old, w_new = np.sqrt(1 - 0.8) * np.ones(3), np.sqrt(0.8) * np.ones(3)
    outs = []
    for tune in (1, 0):
        ctx.set_option(bh.OPT_TUNE_PLACEMENT, tune)
        try:
            sc = bh.SegChains(segs, mu, chol, n, seed=3, mcnext=False, pooled=True)
            info = []
            for i in range(2):
                t, a, b = C.c_int(), C.c_float(), C.c_float()
                ctx.check(ctx.lib.bhip_segchains_placement_info(sc.h, i, C.byref(t), C.byref(a), C.byref(b)))
                info.append(t.value)
            sc.step(w_old, w_new)
            ll, acc, y0 = sc.state()
            X = sc.paths(1, n - 2, 2)[0]
            outs.append((ll, acc, y0, X, info))
            del sc
            torch.cuda.empty_cache()
        finally:
            ctx.set_option(bh.OPT_TUNE_PLACEMENT, 1)
    (lla, acca, y0a, Xa, ia), (llb, accb, y0b, Xb, ib) = outs
    assert np.array_equal(lla, llb) and np.array_equal(acca, accb) and np.array_equal(y0a, y0b) and np.array_equal(Xa, Xb)
    assert acca.sum() > 0
    assert all(1 <= t <= 24 for t in ia) and all(t == 0 for t in ib)
    r = o.smooth_mcmc(refs, mu, chol, w_old, w_new, 3, n - 1)
    assert acca[n - 1] == r["acc"] and np.array_equal(Xa[1], r["X"][1])
