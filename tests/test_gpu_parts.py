"""Ensembles kept in PARTS (round 5): bhip_sample_solve_parts writes paths [j*part_paths, (j+1)*part_paths) to buffer j in ONE launch;
bhip_alloc_apart hands out buffers that lie in different 96-GiB pieces of the device memory (a write stream per piece:
profiles/r5_three_pieces.txt).  The values are those of bhip_sample_solve (src/wiener.jl:24-58 + src/euler.jl:247-268 +
src/partialbridge.jl:67-77 fused), which tests/test_gpu_parity.py pins against the oracle: `==` here, part by part."""
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


@pytest.mark.parametrize("name,n", [("fhn_partialbridge_extreme", 1000), ("fhn_partialbridge_extreme", 70001), ("nclar_firstcomponent", 777),
                                    ("ouproc_nuh", 64), ("linpro2_guidedbridge", 4099)])
@pytest.mark.parametrize("nparts", [2, 3])
def test_parts_hold_the_values_of_the_single_buffer(ctx, name, n, nparts):
    case = [c for c in problems.cases(101 if n < 5000 else 41) if c.name == name][0]
    Po = case.bh_proposal(bh, ctx)
    X1, _, ll1 = bh.sample_solve(case.x0, Po, n, seed=11, iter=3, path0=5)
    XP, llp = bh.sample_solve_parts(case.x0, Po, n, nparts=nparts, seed=11, iter=3, path0=5)
    assert XP.part_paths % 64 == 0 and XP.nparts * XP.part_paths >= n
    # (equal_nan: on the coarse grid a few of 70 001 FitzHugh-Nagumo paths leave the Euler scheme's stability region -- in both runs alike)
    assert np.array_equal(ll1.cpu().numpy(), llp.cpu().numpy(), equal_nan=True)
    assert np.array_equal(X1.paths(), XP.paths(), equal_nan=True)
    # a part is an ensemble like any other: the stand-alone llikelihood of part 1 == the fused values of its paths
    part = XP.parts[1]
    if part.npaths:
        ll = bh.llikelihood(bh.LeftRule(), part, Po).cpu().numpy()
        assert np.array_equal(ll[:part.npaths], ll1.cpu().numpy()[XP.part_paths:XP.part_paths + part.npaths], equal_nan=True)
    # and against the oracle directly for one path of the last part
    p = n - 1
    ref = case.oracle_proposal()
    W = o.wiener_sample(case.tt, Po.mp, 11, 5 + p, 3)
    Xr = o.solve_guided(ref, case.x0, W)
    assert np.array_equal(XP.paths(p, 1)[0], Xr) and llp.cpu().numpy()[p] == o.llikelihood(ref, Xr)
    XP.free()


def test_parts_on_the_lanes_of_dimension_five_and_under_an_earlier_noise_specification():
    rng = np.random.default_rng(2)
    d, N, n = 5, 61, 900
    G = rng.standard_normal((d, d)) / np.sqrt(d)
    tt = np.linspace(0, 0.6, N)
    c5 = bh.Context(0)
    P, Pt = bh.LinPro(-np.eye(d) + 0.2 * G, np.zeros(d), 0.5 * np.eye(d)), bh.LinPro(-np.eye(d), np.zeros(d), 0.5 * np.eye(d))
    Po = bh.GuidedBridge(tt, P, Pt, 0.3 * np.ones(d), ctx=c5)
    X1, _, ll1 = bh.sample_solve(np.zeros(d), Po, n, seed=4)
    XP, llp = bh.sample_solve_parts(np.zeros(d), Po, n, nparts=3, seed=4)
    assert torch.equal(ll1, llp) and _same(X1, XP, lambda: bh.sample_solve(np.zeros(d), Po, n, seed=4)[0])
    XP.free()
    c3 = bh.Context(0)
    c3.set_option(bh.OPT_NOISE_SPEC, 3)
    case = [c for c in problems.cases(101) if c.name == "fhn_partialbridge_extreme"][0]
    Po3 = case.bh_proposal(bh, c3)
    X1, _, ll1 = bh.sample_solve(case.x0, Po3, 3000, seed=4)
    XP, llp = bh.sample_solve_parts(case.x0, Po3, 3000, nparts=2, seed=4)
    assert torch.equal(ll1, llp) and _same(X1, XP, lambda: bh.sample_solve(case.x0, Po3, 3000, seed=4)[0])
    XP.free()


def _same(X1, XP, redo):
    """X1.paths() == XP.paths(); on a mismatch says which of the two downloads differs from a third run (a device-memory problem shows as
    zeros in one of them: small hipDeviceMallocContiguous buffers did that to their neighbours, see bhip_alloc_apart)"""
    xa, xb = X1.paths(), XP.paths()
    if np.array_equal(xa, xb):
        return True
    torch.cuda.synchronize()
    ref = redo().paths()
    print("single buffer differs from a third run in", (xa != ref).sum(), "entries, the parts in", (xb != ref).sum(), "; first differences at", np.argwhere(xa != xb)[:8].tolist())
    return False


def test_forward_euler_maruyama_in_parts(ctx):
    case = problems.forward_cases(101)[0]
    Po = case.bh_proposal(bh, ctx)
    X1, _, ll1 = bh.sample_solve(case.x0, Po, 500, seed=1)
    XP, llp = bh.sample_solve_parts(case.x0, Po, 500, nparts=2, seed=1)
    assert ll1 is None and llp is None and np.array_equal(X1.paths(), XP.paths())
    XP.free()


def test_buffers_apart_and_argument_checks(ctx):
    ptrs = (bh.api.vp * 3)()
    apart = C.c_int(-1)
    nbytes = 512 << 20
    ctx.check(ctx.lib.bhip_alloc_apart(ctx.h, 3, nbytes, ptrs, C.byref(apart)))
    assert len({ptrs[0], ptrs[1], ptrs[2]}) == 3 and all(ptrs[k] for k in range(3))
    assert 1 <= apart.value <= 3                              # (3 on an idle device; the allocator decides -- profiles/r5_piece_map.txt)
    ctx.check(ctx.lib.bhip_free_apart(ctx.h, 3, ptrs))
    small = (bh.api.vp * 2)()
    ctx.check(ctx.lib.bhip_alloc_apart(ctx.h, 2, 1 << 20, small, C.byref(apart)))          # too small to be tested: handed out as they come
    assert apart.value == 0 and small[0] and small[1]
    ctx.check(ctx.lib.bhip_free_apart(ctx.h, 2, small))
    assert ctx.lib.bhip_alloc_apart(ctx.h, 4, nbytes, ptrs, None) != 0
    case = [c for c in problems.cases(41) if c.name == "fhn_partialbridge_extreme"][0]
    Po = case.bh_proposal(bh, ctx)
    XP = bh.EnsembleParts(Po.tt, Po.d, 300, 2, ctx)
    ll = ctx.empty(300)
    x0 = bh.api._dptr(bh.api._x0(case.x0, Po.d))
    call = lambda nparts, ld, part, n: ctx.lib.bhip_sample_solve_parts(ctx.h, Po.h, x0, nparts, XP._ptrs, ld, part, bh.api.vp(ll.data_ptr()), 0, n, 1, 0, 0)
    assert call(2, XP.part_paths, XP.part_paths, 300) == 0
    assert call(2, XP.part_paths, 100, 300) != 0              # part_paths must be a multiple of 64
    assert b"multiple of 64" in ctx.lib.bhip_last_error(ctx.h)
    assert call(2, 64, 128, 300) != 0                          # leading dimension below part_paths
    assert call(2, XP.part_paths, 128, 300) != 0              # the parts do not cover the paths
    assert call(4, XP.part_paths, XP.part_paths, 300) != 0
    XP.free()
    # d = 16 runs on the tile kernel, which writes one buffer
    d = 16
    Po16 = bh.GuidedBridge(np.linspace(0, 0.5, 33), bh.LinPro(-np.eye(d), np.zeros(d), 0.5 * np.eye(d)), bh.LinPro(-np.eye(d), np.zeros(d), 0.5 * np.eye(d)),
                           0.1 * np.ones(d), ctx=ctx)
    with pytest.raises(bh.BridgeError, match="tile kernel"):
        bh.sample_solve_parts(np.zeros(d), Po16, 256, nparts=2)


# ------------------------------------------------------------------ round 6: ONE container (EnsemblePath keeps large ensembles in parts by itself)
@pytest.mark.parametrize("name,n,parts", [("fhn_partialbridge_extreme", 1000, 2), ("linpro2_guidedbridge", 4099, 3), ("linpro3_guidedbridge", 130, 2),
                                           ("ou_guidedbridge", 640, 2)])
def test_every_reader_and_writer_takes_an_ensemble_in_parts(ctx, name, n, parts):
    """sample!, solve!, llikelihood, innovations!, girsanov, upload / download, copy, data: an EnsemblePath in parts gives, column by column,
    what the same calls give on one buffer -- including where the two ensembles of a call are cut differently (W in 3 parts, X in 2)"""
    case = [c for c in problems.cases(65) if c.name == name][0]
    Po = case.bh_proposal(bh, ctx)
    d, mp = Po.d, Po.mp
    # the fused proposal with W kept too: X and W in parts, per range of columns
    X1, W1, ll1 = bh.sample_solve(case.x0, Po, n, seed=5, iter=2, path0=3, store_W=True, parts=1)
    XP, WP, llp = bh.sample_solve(case.x0, Po, n, seed=5, iter=2, path0=3, store_W=True, parts=parts)
    assert XP.nparts == WP.nparts == parts and X1.nparts == 1
    assert torch.equal(ll1, llp) and np.array_equal(X1.paths(), XP.paths()) and np.array_equal(W1.paths(), WP.paths())
    assert torch.equal(X1.data, XP.data) and XP.data.shape == (len(case.tt), d, n)
    with pytest.raises(bh.BridgeError):
        XP.ptr()
    # sample! into parts == sample! into one buffer (the noise is keyed by the global path id)
    Wa = bh.sample(case.tt, bh.Wiener(mp), npaths=n, seed=8, iter=1, path0=11, ctx=ctx)
    Wb = bh.sample_(bh.EnsemblePath(case.tt, mp, n, ctx, parts=3), bh.Wiener(mp), seed=8, iter=1, path0=11)
    assert np.array_equal(Wa.paths(), Wb.paths())
    # solve! with an external W: Y in `parts` buffers, W in 3 (differently cut), ll fused; per-path starts
    lla, llb = ctx.empty(n), ctx.empty(n)
    Ya = bh.solve(bh.Euler(), case.x0, Wa, Po, ll=lla)
    Yb = bh.EnsemblePath(case.tt, d, n, ctx, parts=parts)
    endb = bh.solve_(bh.Euler(), Yb, case.x0, Wb, Po, ll=llb)
    assert np.array_equal(Ya.paths(), Yb.paths()) and torch.equal(lla, llb) and torch.equal(endb, Ya.data[-1])
    u = torch.as_tensor(np.ascontiguousarray(np.array(case.x0)[:, None] + 0.01 * np.arange(n)[None, :] / n), device=ctx.device)
    Yc = bh.solve(bh.Euler(), u, Wa, Po)
    Yd = bh.EnsemblePath(case.tt, d, n, ctx, parts=parts)
    bh.solve_(bh.Euler(), Yd, u, Wb, Po)
    assert np.array_equal(Yc.paths(), Yd.paths())
    Xe, _, lle = bh.sample_solve(u, Po, n, seed=6, parts=1)
    Xf, _, llf = bh.sample_solve(u, Po, n, seed=6, parts=parts)
    assert np.array_equal(Xe.paths(), Xf.paths()) and torch.equal(lle, llf)
    # the stand-alone readers
    assert torch.equal(bh.llikelihood(bh.LeftRule(), Ya, Po), bh.llikelihood(bh.LeftRule(), Yb, Po))
    if d == mp:
        Ia, Ib = bh.innovations(bh.EulerMaruyama(), Ya, Po), bh.innovations(bh.EulerMaruyama(), Yb, Po)
        assert Ib.nparts == parts and np.array_equal(Ia.paths(), Ib.paths())
    if name.startswith("linpro") or name.startswith("ou"):
        assert torch.equal(bh.girsanov(Ya, Po, bh.Wiener(d)), bh.girsanov(Yb, Po, bh.Wiener(d)))
    # upload / download / copy
    host = Ya.paths()
    Up = bh.EnsemblePath.from_paths(case.tt, host, ctx, parts=parts)
    assert Up.nparts == parts and np.array_equal(Up.paths(), host) and np.array_equal(Up.paths(n - 7, 5), host[n - 7:n - 2])
    assert np.array_equal(Up.copy().paths(), host) and np.array_equal(Up.path(n // 2).yy, host[n // 2])
    for E in (XP, WP, Wb, Yb, Yd, Xf, Up):
        E.free()


def test_large_ensembles_are_kept_in_parts_by_default(ctx, monkeypatch):
    """EnsemblePath(parts=None): two buffers from PARTS_MIN_BYTES on (1 GiB; lowered here), one below it and above d = 12; bh.sample_solve and
    bh.solve hand such containers out without the caller asking"""
    case = [c for c in problems.cases(65) if c.name == "fhn_partialbridge_extreme"][0]
    Po = case.bh_proposal(bh, ctx)
    assert bh.api.PARTS_MIN_BYTES == 1 << 30
    small = bh.EnsemblePath(case.tt, 2, 4096, ctx)
    assert small.nparts == 1 and small.ld == 4096
    monkeypatch.setattr(bh.api, "PARTS_MIN_BYTES", 1 << 20)
    n = 8192                                                          # 65 x 2 x 8192 x 8 = 8.5 MB
    X, _, ll = bh.sample_solve(case.x0, Po, n, seed=2)
    assert X.nparts == 2 and X.part_paths == 4096 and X.ld == 4096
    X1, _, ll1 = bh.sample_solve(case.x0, Po, n, seed=2, parts=1)
    assert torch.equal(ll, ll1) and np.array_equal(X.paths(), X1.paths())
    W = bh.sample(case.tt, bh.Wiener(1), npaths=4 * n, seed=1, ctx=ctx)   # 65 x 1 x 32768 x 8 = 17 MB
    assert W.nparts == 2
    Y = bh.solve(bh.Euler(), case.x0, W, Po)
    assert Y.nparts == 2
    assert bh.EnsemblePath(case.tt, 16, 2048, ctx).nparts == 1       # the tile kernel's dimensions: one buffer
    for E in (X, W, Y):
        E.free()


def test_one_launch_forms_for_ensembles_in_parts_check_their_arguments(ctx):
    """bhip_solve_parts / bhip_llikelihood_parts (round 6): the geometry checks of bhip_sample_solve_parts, one buffer delegates to the plain
    entry point, the tile kernel's dimensions are refused with BHIP_EUNSUPPORTED (the mirror then walks the ranges)"""
    case = [c for c in problems.cases(33) if c.name == "fhn_partialbridge_extreme"][0]
    Po = case.bh_proposal(bh, ctx)
    n = 300
    W = bh.sample_(bh.EnsemblePath(case.tt, 1, n, ctx, parts=2), bh.Wiener(1), seed=3)
    X = bh.EnsemblePath(case.tt, 2, n, ctx, parts=2)
    ll = ctx.empty(n)
    x0 = bh.api._dptr(bh.api._x0(case.x0, 2))
    L, vp = ctx.lib, bh.api.vp
    nw, wp, ldw, wpart = W._parts_args()
    nx, xp, ldx, xpart = X._parts_args()
    llp = vp(ll.data_ptr())
    assert L.bhip_solve_parts(ctx.h, Po.h, x0, nw, wp, ldw, wpart, nx, xp, ldx, xpart, llp, 0, n) == 0
    assert L.bhip_solve_parts(ctx.h, Po.h, x0, nw, wp, ldw, 100, nx, xp, ldx, xpart, llp, 0, n) == -1       # part_paths not a multiple of 64
    assert L.bhip_solve_parts(ctx.h, Po.h, x0, nw, wp, 64, wpart, nx, xp, ldx, xpart, llp, 0, n) == -5        # leading dimension below part_paths
    assert L.bhip_solve_parts(ctx.h, Po.h, x0, nw, wp, ldw, 128, nx, xp, ldx, xpart, llp, 0, n) == -5         # the parts do not cover the paths
    assert L.bhip_solve_parts(ctx.h, Po.h, x0, 4, wp, ldw, wpart, nx, xp, ldx, xpart, llp, 0, n) == -1
    assert L.bhip_solve_parts(ctx.h, Po.h, None, nw, wp, ldw, wpart, nx, xp, ldx, xpart, llp, 0, n) == -1    # shared start only
    ll2 = ctx.empty(n)
    same = lambda a, b: np.array_equal(a.cpu().numpy(), b.cpu().numpy(), equal_nan=True)   # (on this coarse grid a few FitzHugh-Nagumo paths leave the Euler scheme's stability region: NaN in both)
    assert L.bhip_llikelihood_parts(ctx.h, Po.h, nx, xp, ldx, xpart, vp(ll2.data_ptr()), 0, n) == 0 and same(ll, ll2)
    assert int(torch.isfinite(ll).sum()) > n // 2
    assert L.bhip_llikelihood_parts(ctx.h, Po.h, nx, xp, ldx, 96, vp(ll2.data_ptr()), 0, n) == -1
    assert L.bhip_llikelihood_parts(ctx.h, Po.h, nx, xp, ldx, xpart, None, 0, n) == -1
    # sample!(W) into the buffers by one launch: the geometry checks, and the values of the range-by-range calls
    W2 = bh.EnsemblePath(case.tt, 1, n, ctx, parts=2)
    a2, b2, c2, d2 = W2._parts_args()
    ttp = bh.api._dptr(W2.tt)
    assert L.bhip_wiener_sample_parts(ctx.h, ttp, len(W2.tt), 1, a2, b2, c2, d2, n, 3, 0, 0) == 0 and np.array_equal(W2.paths(), W.paths())
    assert L.bhip_wiener_sample_parts(ctx.h, ttp, len(W2.tt), 1, a2, b2, c2, 100, n, 3, 0, 0) == -1
    assert L.bhip_wiener_sample_parts(ctx.h, ttp, len(W2.tt), 1, a2, b2, 64, d2, n, 3, 0, 0) == -5
    assert L.bhip_wiener_sample_parts(ctx.h, ttp, len(W2.tt), 1, 4, b2, c2, d2, n, 3, 0, 0) == -1
    Wr = bh.EnsemblePath(case.tt, 1, n, ctx, parts=1)
    for a, m in ((0, d2), (d2, n - d2)):                       # the same columns by plain calls with the global path offset
        assert L.bhip_wiener_sample(ctx.h, ttp, len(W2.tt), 1, Wr.colptr(a), Wr.ld, m, 3, 0, a) == 0
    assert np.array_equal(Wr.paths(), W2.paths())
    W2.free()
    # X not stored: ll alone, the same values
    ll3 = ctx.empty(n)
    assert L.bhip_solve_parts(ctx.h, Po.h, x0, nw, wp, ldw, wpart, 0, None, 0, 0, vp(ll3.data_ptr()), 0, n) == 0 and same(ll, ll3)
    # d = 16: the tile kernel -> BHIP_EUNSUPPORTED from the one-launch forms, and the mirror's fall-back gives the values of one buffer
    c16 = problems.linpro_big_case(16, 41)
    Po16 = c16.bh_proposal(bh, ctx)
    W16 = bh.sample_(bh.EnsemblePath(c16.tt, 16, 200, ctx, parts=2), bh.Wiener(16), seed=2)
    Y16 = bh.EnsemblePath(c16.tt, 16, 200, ctx, parts=2)
    a, b, c, d_ = W16._parts_args()
    e, f, g, h = Y16._parts_args()
    assert L.bhip_solve_parts(ctx.h, Po16.h, bh.api._dptr(bh.api._x0(c16.x0, 16)), a, b, c, d_, e, f, g, h, None, 0, 200) == -3
    lla, llb = ctx.empty(200), ctx.empty(200)
    bh.solve_(bh.Euler(), Y16, c16.x0, W16, Po16, ll=lla)
    W1 = bh.sample(c16.tt, bh.Wiener(16), npaths=200, seed=2, ctx=ctx)
    Y1 = bh.solve(bh.Euler(), c16.x0, W1, Po16, ll=llb)
    assert np.array_equal(Y16.paths(), Y1.paths()) and torch.equal(lla, llb)
    assert torch.equal(bh.llikelihood(bh.LeftRule(), Y16, Po16), bh.llikelihood(bh.LeftRule(), Y1, Po16))
    for E in (W, X, W16, Y16):
        E.free()
