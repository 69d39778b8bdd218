"""Edge cases of the hot path on the GPU (through the C ABI): shortest grids, odd/even step counts
(the time loop is unrolled by two), skip covering everything, single path, padded leading dimensions,
the scripts' real grid length (dt = 1/5000 on T = 2: 10 001 points, partialbridge_fitzhugh.jl:11-14),
and the argument checks."""
import ctypes as C

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


def _fhn_case(tt):
    fpar = [0.1, 0.0, 1.5, 0.8, 0.3]
    return problems.Case("fhn", tt, [-0.5, -0.6], o.MODEL_FHN, fpar, o.AUX_AFFINE, problems.fhn_aux_end(*fpar, 1.1),
                         o.GUIDE_LMMU, 2, 1, m=1, L=[[1.0, 0.0]], v=[1.1], Sigma=[[1e-10]])


@pytest.mark.parametrize("N", [2, 3, 4, 5, 6, 9, 10])
def test_short_grids_and_loop_remainders(ctx, N):
    c = _fhn_case(problems.tau_grid(0.05, N))
    Po, ref = c.bh_proposal(bh, ctx), c.oracle_proposal()
    X, W, ll = bh.sample_solve(c.x0, Po, 67, seed=1, store_W=True)
    Xh, Wh, llh = X.paths(), W.paths(), ll.cpu().numpy()
    for p in (0, 63, 66):
        Wr = o.wiener_sample(c.tt, 1, 1, p, 0)
        Xr = o.solve_guided(ref, c.x0, Wr)
        assert np.array_equal(Wh[p], Wr) and np.array_equal(Xh[p], Xr) and llh[p] == o.llikelihood(ref, Xr)
    # the same through the pCN chain kernel (prefetch window longer than the grid)
    ch = bh.Chains(Po, c.x0, 67, seed=2)
    ch.step(0.5, 4)
    r = o.mcmc(ref, c.x0, 0.5, 4, 2, 66)
    Xc, Wc = ch.paths(66, 1)
    assert ch.acc()[66] == r["acc"] and ch.ll()[66] == r["ll"] and np.array_equal(Xc[0], r["X"]) and np.array_equal(Wc[0], r["W"])
    # stand-alone llikelihood and the inverse of an external-W solve
    assert torch.equal(bh.llikelihood(bh.LeftRule(), X, Po), ll)


def test_skip_semantics_and_single_path(ctx):
    c = _fhn_case(problems.tau_grid(0.5, 41))
    Po, ref = c.bh_proposal(bh, ctx), c.oracle_proposal()
    X, W, ll0 = bh.sample_solve(c.x0, Po, 1, seed=3, store_W=True)
    Xr = o.solve_guided(ref, c.x0, o.wiener_sample(c.tt, 1, 3, 0, 0))
    assert np.array_equal(X.paths()[0], Xr) and float(ll0[0]) == o.llikelihood(ref, Xr)
    for skip in (1, 5, 39, 40, 100):          # skip >= N-1: empty sum, ll = 0 (src/partialbridge.jl:72 loop 1:N-1-skip)
        ll = bh.llikelihood(bh.LeftRule(), X, Po, skip=skip)
        assert float(ll[0]) == o.llikelihood(ref, Xr, skip=min(skip, 40)) and (skip < 40 or float(ll[0]) == 0.0)


def test_padded_leading_dimension_and_raw_pointers(ctx):
    """ld > npaths: the ABI takes raw device pointers with an explicit leading dimension"""
    c = _fhn_case(problems.tau_grid(0.5, 33))
    Po, ref = c.bh_proposal(bh, ctx), c.oracle_proposal()
    P, ld, N = 50, 96, 33
    Wt = torch.full((N, 1, ld), float("nan"), dtype=torch.float64, device=ctx.device)
    Xt = torch.full((N, 2, ld), float("nan"), dtype=torch.float64, device=ctx.device)
    llt = torch.full((ld,), float("nan"), dtype=torch.float64, device=ctx.device)
    x0 = np.array(c.x0)
    ctx.check(ctx.lib.bhip_sample_solve(ctx.h, Po.h, bh.api._dptr(x0), None, bh.api.vp(Wt.data_ptr()), ld, bh.api.vp(Xt.data_ptr()), ld,
                                        bh.api.vp(llt.data_ptr()), 0, P, 7, 0, 0))
    assert bool(torch.isnan(Xt[:, :, P:]).all()) and bool(torch.isnan(Wt[:, :, P:]).all()) and bool(torch.isnan(llt[P:]).all())
    Xr = o.solve_guided(ref, c.x0, o.wiener_sample(c.tt, 1, 7, 49, 0))
    assert np.array_equal(Xt[:, :, 49].cpu().numpy(), Xr) and float(llt[49]) == o.llikelihood(ref, Xr)
    # bhip_solve on the padded W gives the same X
    X2 = torch.full((N, 2, ld), float("nan"), dtype=torch.float64, device=ctx.device)
    ctx.check(ctx.lib.bhip_solve(ctx.h, Po.h, bh.api._dptr(x0), None, bh.api.vp(Wt.data_ptr()), ld, bh.api.vp(X2.data_ptr()), ld, None, 0, P))
    assert torch.equal(X2[:, :, :P], Xt[:, :, :P])
    # argument checks
    lib, h = ctx.lib, ctx.h
    assert lib.bhip_solve(h, Po.h, bh.api._dptr(x0), None, bh.api.vp(Wt.data_ptr()), 10, bh.api.vp(X2.data_ptr()), ld, None, 0, P) == -5
    assert b"leading dimension" in lib.bhip_last_error(h)
    assert lib.bhip_solve(h, Po.h, bh.api._dptr(x0), None, bh.api.vp(Wt.data_ptr()), ld, bh.api.vp(X2.data_ptr()), ld, None, 0, 0) == -1
    assert lib.bhip_solve(h, Po.h, bh.api._dptr(x0), None, None, ld, bh.api.vp(X2.data_ptr()), ld, None, 0, P) == -1
    assert lib.bhip_solve(h, Po.h, None, None, bh.api.vp(Wt.data_ptr()), ld, bh.api.vp(X2.data_ptr()), ld, None, 0, P) == -1
    assert lib.bhip_solve(h, Po.h, bh.api._dptr(x0), None, bh.api.vp(Wt.data_ptr()), ld, bh.api.vp(X2.data_ptr()), ld, None, -1, P) == -1


def test_more_argument_checks(ctx):
    c = _fhn_case(problems.tau_grid(2.0, 129))   # (explicit Euler on much coarser grids lets the odd FitzHugh-Nagumo path blow up)
    Po = c.bh_proposal(bh, ctx)
    lib, h = ctx.lib, ctx.h
    ch = bh.Chains(Po, c.x0, 64, seed=1)
    for rho in (1.5, -1.01, float("nan")):
        with pytest.raises(bh.BridgeError, match="rho"):
            ch.step(rho, 1)
    assert np.isfinite(ch.ll()).all()
    ch.step(1.0, 1)                       # rho = 1: the proposal equals the current W, always accepted
    assert np.array_equal(ch.acc(), np.ones(64, dtype=np.int64))
    # the RNG counter holds the global path id in 32 bits
    with pytest.raises(bh.BridgeError, match="32-bit"):
        bh.Chains(Po, c.x0, 64, seed=1, path0=2 ** 32 - 10)
    with pytest.raises(bh.BridgeError, match="32-bit"):
        bh.sample_solve(c.x0, Po, 64, seed=1, path0=2 ** 32 - 63)
    X, _, _ = bh.sample_solve(c.x0, Po, 64, seed=1, path0=2 ** 32 - 64)      # the last admissible shard
    assert bool(torch.isfinite(X.data).all())
    # a proposal is tied to the context that created it
    other = bh.Context(0)
    ll = ctx.empty(64)
    assert lib.bhip_llikelihood(other.h, Po.h, X.ptr(), 64, bh.api.vp(ll.data_ptr()), 0, 64) == -1
    assert b"another context" in lib.bhip_last_error(other.h)
    # parameter vector missing
    hh = bh.api.vp()
    assert lib.bhip_proposal_create(h, bh.api._dptr(np.linspace(0, 1, 5)), 5, o.MODEL_OU, 1, None, 2, C.byref(hh)) == -1


def test_script_grid_10001_points(ctx):
    """the paper scripts' grid: dt = 1/5000, T = 2, time-changed (partialbridge_fitzhugh.jl:11-14)"""
    T, dt = 2.0, 1 / 5000
    s = np.arange(0, 10001) * dt
    tt = s * (2 - s / T)
    c = _fhn_case(tt)
    Po, ref = c.bh_proposal(bh, ctx), c.oracle_proposal()
    g = c.oracle_guide()
    assert np.array_equal(Po.L, g["L"]) and np.array_equal(Po.M, g["M"]) and np.array_equal(Po.mu, g["mu"])
    ch = bh.Chains(Po, c.x0, 128, seed=44)
    ch.step(0.9, 5)
    r = o.mcmc(ref, c.x0, 0.9, 5, 44, 100)
    X, W = ch.paths(100, 1)
    assert ch.acc()[100] == r["acc"] and ch.ll()[100] == r["ll"] and np.array_equal(X[0], r["X"]) and np.array_equal(W[0], r["W"])
    assert abs(X[0, -1, 0] - 1.1) < 1e-3                                           # the bridge ends at the observation


def test_handles_release_their_device_memory(ctx):
    """proposals, chains and ensembles are created and dropped repeatedly: the free device memory must come back"""
    import gc
    c = _fhn_case(problems.tau_grid(2.0, 257))

    def cycle(rep):
        Po = c.bh_proposal(bh, ctx)
        ch = bh.Chains(Po, c.x0, 4096, seed=rep)
        ch.step(0.9, 1)
        ch.pathstats()
        ch.paths(0, 8)
        X, _, _ = bh.sample_solve(c.x0, Po, 4096, seed=rep)
        del ch, Po, X

    def free():
        gc.collect()
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        return torch.cuda.mem_get_info()[0]

    cycle(0)        # first use loads the kernels' code objects and the runtime's pools (~150 MB, once)
    free0 = free()
    for rep in range(40):
        cycle(rep)
    free1 = free()
    assert free0 - free1 < 64 << 20, (free0, free1)          # each repetition allocates > 50 MB: a leak would show as GBs
