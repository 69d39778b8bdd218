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
