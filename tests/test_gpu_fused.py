"""BHIP_OPT_FUSED_ARITHMETIC: the d <= 3 kernels from a second build of the same source with a*b + c contracted (-ffp-contract=fast).
The default build is compared with the CPU restatement bit for bit (tests/test_gpu_parity.py); this option trades that for ~12 %
fewer vector instructions and is held to the stated fp64 tolerance instead: 1e-9 on paths, 1e-8 on log-likelihoods."""
import numpy as np
import pytest
import torch

import bridgehip as bh
import oracle as o
import problems

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fctx():
    c = bh.Context(0)
    c.set_option(bh.OPT_FUSED_ARITHMETIC, 1)
    return c


@pytest.mark.parametrize("case", problems.cases(129), ids=lambda c: c.name)
def test_fused_paths_and_loglikelihoods_within_the_stated_tolerance(fctx, case):
    Po, ref = case.bh_proposal(bh, fctx), case.oracle_proposal()
    for P in (70, 2100):                                          # wave-specialised kernel, then ... (both are <= 98 304: see below)
        X, W, ll = bh.sample_solve(case.x0, Po, P, seed=12, iter=1, path0=7, store_W=True)
        Xh, Wh, llh = X.paths(), W.paths(), ll.cpu().numpy()
        for p in (0, 63, P - 1):
            Wr = o.wiener_sample(case.tt, case.mp, 12, 7 + p, 1)
            assert np.abs(Wh[p] - Wr).max() <= 1e-13 * (1 + np.abs(Wr).max())          # the cumulation may fuse its last operation
            Xr = o.solve_guided(ref, case.x0, Wh[p])                                   # the arithmetic, on the kernel's own W
            assert np.isfinite(Xh[p]).all() or not np.isfinite(Xr).all()
            if np.isfinite(Xr).all():
                assert np.abs(Xh[p] - Xr).max() <= 1e-9 * (1 + np.abs(Xr).max()), (case.name, p)
                lr = o.llikelihood(ref, Xr)
                assert abs(llh[p] - lr) <= 1e-8 * (1 + abs(lr)), (case.name, p)
        # external W and the stand-alone llikelihood run fused too
        ll2 = fctx.empty(P)
        X2 = bh.solve(bh.Euler(), case.x0, W, Po, ll=ll2)
        assert float((X2.data - X.data).abs().max()) <= 1e-9 * (1 + float(X.data.abs().max()))
        ll3 = bh.llikelihood(bh.LeftRule(), X, Po)
        fin = torch.isfinite(ll)
        assert float((ll3 - ll)[fin].abs().max()) <= 1e-8 * (1 + float(ll[fin].abs().max()))


def test_fused_kernels_differ_from_the_exact_ones_only_in_rounding_and_chains_agree_in_law():
    ectx, fctx = bh.Context(0), bh.Context(0)
    fctx.set_option(bh.OPT_FUSED_ARITHMETIC, 1)
    case = [c for c in problems.cases(257) if c.name == "fhn_partialbridge_extreme"][0]
    n, iters = 20000, 12
    res = []
    for ctx in (ectx, fctx):
        Po = case.bh_proposal(bh, ctx)
        ch = bh.Chains(Po, case.x0, n, seed=3)
        ch.step(0.9, iters)
        res.append((ch.ll().copy(), ch.acc().copy(), ch.paths(0, 64)[0]))
    (lle, acce, Xe), (llf, accf, Xf) = res
    assert not np.array_equal(lle, llf)                                   # it IS another build ...
    same = acce == accf                                                    # ... whose chains take the same decisions except where
    assert same.mean() > 0.995                                             #     llo - ll - log U sits within rounding of zero
    assert np.abs(lle[same] - llf[same]).max() <= 1e-8 * (1 + np.abs(lle[same]).max())
    assert np.abs(Xe[same[:64]] - Xf[same[:64]]).max() <= 1e-9 * (1 + np.abs(Xe).max())
    assert abs(acce.mean() - accf.mean()) < 0.02 * iters


def test_K9_of_the_regrouped_step_at_a_million_paths(fctx):
    """C2 under BHIP_OPT_FUSED_ARITHMETIC runs the REGROUPED step at d = 1 (GUIDE_QF rows, one dependent FMA per step; round 6).  K9
    (test/guip.jl:245-274) at 2^20 paths on the 1001-point grid: the importance weights have mean 1 within 5 standard errors + the Euler
    scheme's O(dt) allowance, and -- the noise being the same stream -- every path's log-likelihood agrees with the exact build's to the
    stated 1e-8, on the one-path-per-lane kernel (2^20 paths) and on the wave-specialised one (65 536 paths)."""
    import math
    c = [k for k in problems.cases(1001) if k.name == "ou_guidedbridge"][0]
    ectx = bh.Context(0)
    beta, a, T, u, v = 0.8, 0.7, 2.0, float(c.x0[0]), float(c.v[0])
    K = a / (2 * beta) * (1 - math.exp(-2 * beta * T))
    lp = -0.5 * ((v - u * math.exp(-beta * T)) ** 2 / K + math.log(K) + math.log(2 * math.pi))
    Pof, Poe = c.bh_proposal(bh, fctx), c.bh_proposal(bh, ectx)
    for P in (1 << 20, 65536):
        _, _, llf = bh.sample_solve(c.x0, Pof, P, seed=2025, store_X=False)
        _, _, lle = bh.sample_solve(c.x0, Poe, P, seed=2025, store_X=False)
        assert bool(torch.isfinite(llf).all())
        assert float(((llf - lle).abs() / (1 + lle.abs())).max()) <= 1e-8
        assert not torch.equal(llf, lle)                                  # (it IS another arithmetic)
        w = torch.exp(llf + (bh.lptilde(Pof, c.x0) - lp))
        assert abs(float(w.mean()) - 1.0) < 5 * math.sqrt(float(w.var()) / P) + 5e-3, (P, float(w.mean()))
