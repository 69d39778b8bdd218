"""GPU tests of the noise specifications (bhip_rng.h; replaces the reference's randn, src/wiener.jl:24-58).

(1) The selectable streams bhip-philox-v3 (BHIP_OPT_NOISE_SPEC = 3: two Box-Muller pairs of 40 + 24 bits per Philox call, the default
    of rounds 3 and 4) and bhip-philox-v2 (= 2: one pair of 53 + 53 bits per call) against their oracle twins (bo_set_noise_spec,
    pinned on CPU by goldens v4 and v3 from their seeds): every kernel family that draws normals.
(2) The JOINT law of consecutive normals under every specification.  v3 / v2: the marginal tests (K13, tests/test_gpu_fullsize.py)
    cannot see a defect in how a Box-Muller pair is formed -- the angle lives on 2^24 rays, the two pairs of a call share its 128 bits.
    The default v4 (one normal per 32-bit word through the inverse distribution function): the four normals of a call are four words
    of one Philox output.  So: chi-square of the pair angle, of the squared radius against Exp(1/2), of angle x angle and radius x
    angle within a call, on 1.3e8 pairs; the realised quadratic variation of 1000-step Wiener paths against its chi-square law.
(3) v4 and v3 against v2 on path functionals at 10^6 paths: the importance weights of K9 (test/guip.jl:245-274) -- two-sample
    comparisons, so the Euler scheme's own O(dt) bias (identical under all streams) does not enter.
(4) The marginal of v4 where an inverse-distribution-function generator could fail: tail frequencies at |z| > 4.5 and 5.5 on 2.6e9
    in-kernel draws, the reach of the tail, a fine chi-square of the marginal; and (CPU, tests/test_oracle.py) the Kolmogorov distance
    read off the table.
"""
import math

import numpy as np
import pytest
import torch

import bridgehip as bh
import oracle as o
import problems

pytestmark = pytest.mark.gpu
SEED = 123


@pytest.fixture(scope="module")
def ctx():
    return bh.default_context(0)


@pytest.fixture(scope="module")
def ctx2():
    c = bh.Context(0)
    c.set_option(bh.OPT_NOISE_SPEC, 2)
    return c


@pytest.fixture(scope="module")
def ctx3():
    c = bh.Context(0)
    c.set_option(bh.OPT_NOISE_SPEC, 3)
    return c


@pytest.fixture
def ctxs(request, ctx2, ctx3):
    """the context of the non-default specification `spec` (indirect parametrisation)"""
    return {2: ctx2, 3: ctx3}[request.param], request.param


OLD_SPECS = pytest.mark.parametrize("ctxs", [3, 2], indirect=True, ids=["v3", "v2"])


def _case(name, N=151):
    return [c for c in problems.cases(N) + problems.forward_cases(N) if c.name == name][0]


# ------------------------------------------------------------------------------------------------ (1) the full-resolution stream
@OLD_SPECS
@pytest.mark.parametrize("mp", [1, 2, 3, 5])
def test_v2_wiener_sample_bit_exact(ctxs, mp):
    ctx2, spec = ctxs
    tt = problems.tau_grid(2.0, 203)
    W = bh.sample(tt, bh.Wiener(mp), npaths=130, seed=SEED, iter=2, path0=7, ctx=ctx2).paths()
    with o.noise_spec(spec):
        for p in (0, 63, 64, 129):
            assert np.array_equal(W[p], o.wiener_sample(tt, mp, SEED, 7 + p, 2)), (mp, p)
    assert not np.array_equal(W[0], o.wiener_sample(tt, mp, SEED, 7, 2))      # ... and it is not the default stream


@pytest.mark.parametrize("name", ["fhn_partialbridge_extreme", "ou_guidedbridge", "nclar_firstcomponent", "linpro2_guidedbridge", "linpro3_guidedbridge",
                                  "fhn2_nuh_full"])
@pytest.mark.parametrize("P", [200, 99000])
@OLD_SPECS
def test_v2_fresh_proposals_match_oracle(ctxs, name, P):
    """bhip_sample_solve under spec 2 / 3: the wave-specialised kernel (P = 200: k_pc) and the one-lane kernel (P = 99 000 > 98 304: k_paths)"""
    ctx2, spec = ctxs
    c = _case(name)
    Po, ref = c.bh_proposal(bh, ctx2), c.oracle_proposal()
    X, W, ll = bh.sample_solve(c.x0, Po, P, seed=SEED, iter=1, path0=11, store_W=True)
    llh = ll.cpu().numpy()
    with o.noise_spec(spec):
        for p in (0, 1, 65, P - 1):
            Wr = o.wiener_sample(c.tt, c.mp, SEED, 11 + p, 1)
            assert np.array_equal(W.paths(p, 1)[0], Wr), (name, p)
            Xr = o.solve_guided(ref, c.x0, Wr)
            if c.exact:
                assert np.array_equal(X.paths(p, 1)[0], Xr) and llh[p] == o.llikelihood(ref, Xr), (name, p)
            else:
                assert np.abs(X.paths(p, 1)[0] - Xr).max() <= 1e-9 * (1 + np.abs(Xr).max())


@pytest.mark.parametrize("wave_specialised", [1, 0])
@pytest.mark.parametrize("name", ["fhn_partialbridge_extreme", "ou_guidedbridge", "linpro3_guidedbridge", "nclar_full"])
@OLD_SPECS
def test_v2_chains_match_oracle(ctxs, name, wave_specialised):
    """pCN chains under spec 2 / 3 on the wave-specialised kernel (BHIP_OPT_WAVE_SPECIALISED = 0 changes nothing under these
    specifications: the one-lane twin k_chain_lines holds the default stream only): decisions, log-likelihoods, Wiener states and
    paths of the oracle's chain under the same specification"""
    ctx2, spec = ctxs
    c = _case(name)
    ctx2.set_option(bh.OPT_WAVE_SPECIALISED, wave_specialised)
    try:
        ch = bh.Chains(c.bh_proposal(bh, ctx2), c.x0, 70, seed=SEED, path0=5)
        ch.step(c.rho, 12)
        X, W = ch.paths()
        ll, acc = ch.ll(), ch.acc()
    finally:
        ctx2.set_option(bh.OPT_WAVE_SPECIALISED, 1)
    ref = c.oracle_proposal()
    with o.noise_spec(spec):
        for p in (0, 63, 64, 69):
            r = o.mcmc(ref, c.x0, c.rho, 12, SEED, 5 + p)
            if c.exact:
                assert acc[p] == r["acc"] and ll[p] == r["ll"], (name, p)
                assert np.array_equal(W[p], r["W"]) and np.array_equal(X[p], r["X"])
            elif acc[p] == r["acc"]:
                assert abs(ll[p] - r["ll"]) <= 1e-8 * (1 + abs(r["ll"])) and np.abs(W[p] - r["W"]).max() <= 1e-12
    r4 = o.mcmc(ref, c.x0, c.rho, 12, SEED, 5)     # ... and it is not the default stream
    assert not np.array_equal(W[0], r4["W"])


@pytest.mark.parametrize("d", [5, 16])
@OLD_SPECS
def test_v2_large_dimensions(ctxs, d):
    """d = 5: one path per lane (k_paths<MLinPro<5>>; chains on the tile kernel under these specifications); d = 16: the MFMA tile kernel
    (a quad per lane and pass, 4 x 4 exchange).  Wiener paths bit-exact under spec 2 / 3, paths / ll at the large-d tolerance, chain
    decisions those of the oracle."""
    ctx2, spec = ctxs
    c = problems.linpro_big_case(d, 61)
    Po, ref = c.bh_proposal(bh, ctx2), c.oracle_proposal()
    X, W, ll = bh.sample_solve(c.x0, Po, 90, seed=6, iter=2, path0=100, store_W=True)
    Xh, Wh, llh = X.paths(), W.paths(), ll.cpu().numpy()
    ch = bh.Chains(Po, c.x0, 40, seed=8)
    ch.step(0.95, 6)
    acc, llc = ch.acc(), ch.ll()
    with o.noise_spec(spec):
        for p in (0, 17, 89):
            Wr = o.wiener_sample(c.tt, d, 6, 100 + p, 2)
            assert np.array_equal(Wh[p], Wr), (d, p)
            Xr = o.solve_guided(ref, c.x0, Wr)
            assert np.abs(Xh[p] - Xr).max() <= 1e-9 * (1 + np.abs(Xr).max())
            assert abs(llh[p] - o.llikelihood(ref, Xr)) <= 1e-8 * (1 + abs(llh[p]))
        same = 0
        for p in (0, 13, 39):
            r = o.mcmc(ref, c.x0, 0.95, 6, 8, p)
            if acc[p] == r["acc"]:
                same += 1
                assert abs(llc[p] - r["ll"]) <= 1e-8 * (1 + abs(r["ll"]))
        assert same >= 2


@OLD_SPECS
def test_v2_ensembles_keep_their_specification(ctx, ctxs):
    ctx2, spec = ctxs
    c = _case("fhn_partialbridge_extreme")
    a = bh.Chains(c.bh_proposal(bh, ctx2), c.x0, 64, seed=3)
    a.step(0.9, 3)
    state = a.save()
    # an ensemble of the default specification refuses the state ...
    b = bh.Chains(c.bh_proposal(bh, ctx), c.x0, 64, seed=3)
    with pytest.raises(bh.BridgeError, match="noise specification"):
        b.load(state)
    # ... one of the same specification resumes it exactly
    a2 = bh.Chains(c.bh_proposal(bh, ctx2), c.x0, 64, seed=3)
    a2.load(state)
    a.step(0.9, 4)
    a2.step(0.9, 4)
    assert np.array_equal(a.ll(), a2.ll()) and np.array_equal(a.acc(), a2.acc())
    # switching the context's option under a live ensemble is refused, not silently mixed
    c3 = bh.Context(0)
    e = bh.Chains(c.bh_proposal(bh, c3), c.x0, 64, seed=3)
    c3.set_option(bh.OPT_NOISE_SPEC, spec)
    with pytest.raises(bh.BridgeError, match="noise specification"):
        e.step(0.9, 1)
    c3.set_option(bh.OPT_NOISE_SPEC, 4)
    e.step(0.9, 1)
    with pytest.raises(bh.BridgeError):
        c3.set_option(bh.OPT_NOISE_SPEC, 5)


# ------------------------------------------------------------------------------------------------ (2) the joint law of a pair
def _chi2_ok(counts, expected, k=5.0):
    """Pearson statistic of `counts` against the constant `expected`: within k standard deviations of its mean B - 1"""
    B = counts.numel()
    stat = float(((counts.double() - expected) ** 2).sum() / expected)
    return abs(stat - (B - 1)) < k * math.sqrt(2 * (B - 1)), stat


def _normals(c, P=262144, N=1001, seed=77):
    """262 144 x 1000 in-kernel normals: increments of a Wiener ensemble on a unit grid; z[i, p] = normal i of path p"""
    W = bh.sample(np.arange(N, dtype=np.float64), bh.Wiener(1), npaths=P, seed=seed, ctx=c)
    return W.data[1:, 0, :] - W.data[:-1, 0, :]


@pytest.mark.parametrize("spec", [4, 3, 2])
def test_joint_law_of_the_box_muller_pairs(ctx, ctx2, ctx3, spec):
    """(v4: no Box-Muller -- consecutive normals 2h, 2h + 1 are two words of one Philox output; the pair must still be the
    rotation-invariant planar Gaussian)"""
    z = _normals({4: ctx, 3: ctx3, 2: ctx2}[spec])
    z0, z1 = z[0::2], z[1::2]                       # pair h = normals 2h (radius * cos), 2h + 1 (radius * sin)
    npairs = z0.numel()
    assert npairs == 500 * 262144
    # the angle: uniform on [0, 1) -- under v3 on the grid k 2^-24 (256 grid points per bin at 2^16 bins; bins shifted by half a grid
    # step so that a grid point never sits on an edge)
    u = torch.remainder(torch.atan2(z1, z0) / (2 * math.pi) + 2.0 ** -25, 1.0)
    for B in (1 << 12, 1 << 16):
        cnt = torch.bincount((u * B).long().clamp_(0, B - 1).flatten(), minlength=B)
        ok, stat = _chi2_ok(cnt, npairs / B)
        assert ok, ("angle", spec, B, stat)
    # the squared radius: z0^2 + z1^2 ~ Exp(1/2), i.e. exp(-s/2) uniform on (0, 1]
    s = z0 * z0 + z1 * z1
    v = torch.exp(-0.5 * s)
    for B in (1 << 12, 1 << 16):
        cnt = torch.bincount((v * B).long().clamp_(0, B - 1).flatten(), minlength=B)
        ok, stat = _chi2_ok(cnt, npairs / B)
        assert ok, ("radius", spec, B, stat)
    # tail of the radius on its own scale: P(s > t) = exp(-t/2)
    for t in (20.0, 30.0):
        n_t, e_t = float((s > t).sum()), npairs * math.exp(-t / 2)
        assert abs(n_t - e_t) < 5 * math.sqrt(e_t) + 1, (spec, t, n_t, e_t)
    # radius x angle of one pair (v3: both from the same 64 bits), 64 x 64 cells
    cell = (v * 64).long().clamp_(0, 63) * 64 + (u * 64).long().clamp_(0, 63)
    ok, stat = _chi2_ok(torch.bincount(cell.flatten(), minlength=4096), npairs / 4096)
    assert ok, ("radius x angle", spec, stat)
    # the two pairs of a v3 call (pairs 2q, 2q + 1 = the two halves of one Philox output; v2: consecutive calls): angle x angle,
    # radius x radius, 64 x 64 cells each
    ua, ub, va, vb = u[0::2], u[1::2], v[0::2], v[1::2]
    for a, b, what in ((ua, ub, "angle x angle"), (va, vb, "radius x radius"), (va, ub, "radius x other angle")):
        cell = (a * 64).long().clamp_(0, 63) * 64 + (b * 64).long().clamp_(0, 63)
        ok, stat = _chi2_ok(torch.bincount(cell.flatten(), minlength=4096), a.numel() / 4096)
        assert ok, (what, spec, stat)
    # rotation invariance seen through a functional the angle grid could bias: E[z0^2 z1^2] = 1, E[z0 z1] = 0, E[z0^3 z1] = 0
    n = float(npairs)
    assert abs(float((z0 * z1).mean())) < 5 / math.sqrt(n)
    assert abs(float((z0 * z0 * z1 * z1).mean()) - 1.0) < 5 * math.sqrt(8.0 / n)      # Var(z0^2 z1^2) = 9 - 1
    assert abs(float((z0 ** 3 * z1).mean())) < 5 * math.sqrt(15.0 / n)


@pytest.mark.parametrize("spec", [4, 3, 2])
def test_realised_quadratic_variation_of_wiener_paths(ctx, ctx2, ctx3, spec):
    """[W]_T over 1000 unit steps is chi-square with 1000 degrees of freedom, path by path: mean 1000, variance 2000, and the
    empirical distribution over 262 144 paths in 64 equiprobable cells"""
    from scipy.stats import chi2
    z = _normals({4: ctx, 3: ctx3, 2: ctx2}[spec], seed=78)
    qv = (z * z).sum(0)
    P = qv.numel()
    assert abs(float(qv.mean()) - 1000.0) < 5 * math.sqrt(2000.0 / P)
    # Var of the sample variance of chi2_k: (kappa - 1) sigma^4 / P with excess kurtosis 12 / k
    assert abs(float(qv.var()) - 2000.0) < 5 * 2000.0 * math.sqrt((2.0 + 12.0 / 1000.0) / P)
    edges = torch.tensor(chi2.ppf(np.arange(1, 64) / 64.0, 1000), dtype=torch.float64, device=qv.device)
    cnt = torch.bincount(torch.bucketize(qv, edges), minlength=64)
    ok, stat = _chi2_ok(cnt, P / 64.0)
    assert ok, (spec, stat)
    # the same for the second half of every path alone (steps 500..999): the increments of one path do not share angle rays
    qv2 = (z[500:] * z[500:]).sum(0)
    assert abs(float(qv2.mean()) - 500.0) < 5 * math.sqrt(1000.0 / P)
    assert abs(float(((qv - qv2) * qv2).mean()) - float((qv - qv2).mean()) * float(qv2.mean())) < 5 * math.sqrt(1000.0 * 1000.0 / P)


# ------------------------------------------------------------------------------------------------ (3) v3 against v2 on path functionals
def test_K9_importance_weights_v4_and_v3_against_v2_at_a_million_paths(ctx, ctx2, ctx3):
    """K9 (test/guip.jl:245-274): w = exp(ll) ptilde / p has mean 1 in continuous time; on the 1001-point grid the Euler scheme adds
    its own O(dt) bias, the same under every stream.  So: (a) all means within 5 standard errors + a 5e-3 allowance for that bias of
    1, (b) the TWO-SAMPLE comparisons v4 - v2 and v3 - v2 of the mean weight, of the mean log-likelihood and of its variance within 5
    standard errors of the difference -- 2^20 paths each, which would show a relative defect of 1e-3 in these functionals."""
    c = _case("ou_guidedbridge", 1001)
    P = 1 << 20
    beta, a, T, u, v = 0.8, 0.7, 2.0, float(c.x0[0]), float(c.v[0])
    K = a / (2 * beta) * (1 - math.exp(-2 * beta * T))
    lp = -0.5 * ((v - u * math.exp(-beta * T)) ** 2 / K + math.log(K) + math.log(2 * math.pi))
    res = {}
    for spec, cx in ((4, ctx), (3, ctx3), (2, ctx2)):
        Po = c.bh_proposal(bh, cx)
        _, _, ll = bh.sample_solve(c.x0, Po, P, seed=2024, store_X=False)
        w = torch.exp(ll + (bh.lptilde(Po, c.x0) - lp))
        res[spec] = (float(w.mean()), float(w.var()), float(ll.mean()), float(ll.var()), float(((ll - ll.mean()) ** 4).mean()))
        assert abs(res[spec][0] - 1.0) < 5 * math.sqrt(res[spec][1] / P) + 5e-3, (spec, res[spec])
    m2, v2, l2, s2, k2 = res[2]
    for spec in (4, 3):
        m3, v3, l3, s3, k3 = res[spec]
        assert abs(m3 - m2) < 5 * math.sqrt((v3 + v2) / P), (spec, m3, m2)
        assert abs(l3 - l2) < 5 * math.sqrt((s3 + s2) / P), (spec, l3, l2)
        se_var = math.sqrt(((k3 - s3 * s3) + (k2 - s2 * s2)) / P)
        assert abs(s3 - s2) < 5 * se_var, (spec, s3, s2, se_var)


# ------------------------------------------------------------------------------------------------ (4) the marginal of v4
def test_v4_tail_frequencies_and_reach_on_2_6e9_draws(ctx):
    """An inverse-distribution-function normal fails, if it fails, in the tails (a wrong segment, a sign slip in a far octave) -- where
    a chi-square over equiprobable cells has no power.  Ten launches of 262 144 x 1000 in-kernel normals = 2.6e9 draws: the counts
    beyond 4.5 and 5.5 (two-sided: 1.8e4 and 99 expected) within 5 standard deviations, both signs separately, the largest |z| where it
    must be (P(nothing beyond 5.7 in 2.6e9 draws) = 3e-14; nothing can lie beyond -Phi^-1(2^-33) = 6.338), and a 2^16-cell chi-square
    of the marginal through the exact distribution function."""
    from scipy.stats import norm
    n, zmax = 0, 0.0
    cnt = {(t, sg): 0 for t in (4.5, 5.5) for sg in (-1, 1)}
    B = 1 << 16
    cells = torch.zeros(B, dtype=torch.long, device="cuda")
    for seed in range(1000, 1010):
        z = _normals(ctx, seed=seed)
        n += z.numel()
        zmax = max(zmax, float(z.abs().max()))
        for t in (4.5, 5.5):
            cnt[(t, 1)] += int((z > t).sum())
            cnt[(t, -1)] += int((z < -t).sum())
        if seed < 1002:   # 5.2e8 draws into 65 536 equiprobable cells
            u = torch.special.ndtr(z)
            cells += torch.bincount((u * B).long().clamp_(0, B - 1).flatten(), minlength=B)
        del z
    assert n == 10 * 1000 * 262144
    for (t, sg), k in cnt.items():
        e = n * norm.sf(t)
        assert abs(k - e) < 5 * math.sqrt(e) + 1, (t, sg, k, e)
    assert 5.7 < zmax < 6.3381, zmax
    ok, stat = _chi2_ok(cells, 2 * 1000 * 262144 / B)
    assert ok, stat
