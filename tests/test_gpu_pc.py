"""GPU tests of the wave-specialised (producer/consumer) kernels, bridge.jl_amd/csrc/bhip_pc_kernel.h.

The producer wave draws the Wiener noise / does the pCN mix and the chain-state traffic, the consumer wave runs the
Euler recurrence + log-likelihood of src/euler.jl:247-268, src/partialbridge.jl:67-77.  They must give, bit for bit,
what the one-lane-does-everything kernels give (which the parity tests compare with the oracle): same Wiener paths,
paths, log-likelihoods, accept decisions -- for every test problem with noise dimension 1, 2 or 3, ragged ensemble sizes
(tails of the 64-path workgroup), grids that are not multiples of the 16-value chunk, with and without the stores.
The default context runs the wave-specialised kernels, so the oracle parity tests of test_gpu_parity.py already go
through them; here both variants are run side by side through the C ABI option BHIP_OPT_WAVE_SPECIALISED.
"""
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


def both(ctx, fn):
    out = []
    for v in (1, 0):
        ctx.set_option(bh.OPT_WAVE_SPECIALISED, v)
        try:
            out.append(fn())
        finally:
            ctx.set_option(bh.OPT_WAVE_SPECIALISED, 1)
    return out


CASES = [c for c in problems.cases(143) + problems.forward_cases(143) if c.mp in (1, 2, 3)]   # m' = 3: lines padded to 4 components


@pytest.mark.parametrize("case", CASES, ids=lambda c: c.name)
@pytest.mark.parametrize("P", [1, 63, 64, 200])
def test_fresh_proposals_equal_monolithic(ctx, case, P):
    Po = case.bh_proposal(bh, ctx)

    def run():
        X, W, ll = bh.sample_solve(case.x0, Po, P, seed=77, iter=5, path0=123, store_W=True)
        return X.paths(), W.paths(), None if ll is None else ll.cpu().numpy()
    (Xa, Wa, la), (Xb, Wb, lb) = both(ctx, run)
    assert np.array_equal(Wa, Wb) and np.array_equal(Xa, Xb)
    assert (la is None and lb is None) or np.array_equal(la, lb)
    # and against the oracle directly (chain p of the ensemble = global path id 123 + p)
    p = P - 1
    W_ref = o.wiener_sample(case.tt, case.mp, 77, 123 + p, 5)
    assert np.array_equal(Wa[p], W_ref)


@pytest.mark.parametrize("case", [c for c in CASES if c.kind != o.GUIDE_NONE], ids=lambda c: c.name)
def test_pcn_chains_equal_monolithic(ctx, case):
    Po = case.bh_proposal(bh, ctx)

    def run():
        ch = bh.Chains(Po, case.x0, 150, seed=31, path0=7)
        ch.step(case.rho, 6)
        X, W = ch.paths(0, 150)
        return X, W, ch.ll(), ch.acc(), ch.proposal_X().cpu().numpy(), ch.stats().cpu().numpy()
    a, b = both(ctx, run)
    for u, v in zip(a, b):
        assert np.array_equal(u, v)
    assert 0 < a[3].sum() < 6 * 150


@pytest.mark.parametrize("N", [2, 3, 4, 5, 6, 9, 16, 17, 18, 33, 129])
def test_short_and_ragged_grids(ctx, N):
    """grids around the chunk size: 16 values per chunk = 16 grid points (m' = 1) / 8 (m' = 2) / 4 (m' = 3, padded lines)"""
    # (FitzHugh-Nagumo is stiff: on grids this coarse its Euler scheme overflows, so the d = 1 bridge stands in below 129 points)
    for name in (("fhn_partialbridge_extreme" if N >= 129 else "ou_guidedbridge"), "linpro2_guidedbridge", "linpro3_guidedbridge", "linpro3_partial_m2"):
        case = [c for c in problems.cases(N) if c.name == name][0]
        Po = case.bh_proposal(bh, ctx)

        def run():
            X, W, ll = bh.sample_solve(case.x0, Po, 70, seed=3, store_W=True)
            ch = bh.Chains(Po, case.x0, 70, seed=3)
            ch.step(0.8, 3)
            Xc, Wc = ch.paths(0, 70)
            return X.paths(), W.paths(), ll.cpu().numpy(), Xc, Wc, ch.ll(), ch.acc()
        a, b = both(ctx, run)
        for u, v in zip(a, b):
            assert np.array_equal(u, v), (name, N)
        ref = case.oracle_proposal()
        r = o.mcmc(ref, case.x0, 0.8, 3, 3, 69)
        assert np.array_equal(a[3][69], r["X"]) and np.array_equal(a[4][69], r["W"]) and a[5][69] == r["ll"] and a[6][69] == r["acc"]


def test_no_store_variants_and_per_path_starts(ctx):
    """log-likelihood only (no X, no W store) and per-path starting points x0_dev on the wave-specialised kernel"""
    case = [c for c in problems.cases(201) if c.name == "fhn_partialbridge_first"][0]
    Po = case.bh_proposal(bh, ctx)
    P = 130
    rng = np.random.default_rng(0)
    x0s = case.x0[None, :] + 0.05 * rng.standard_normal((P, 2))
    x0_dev = torch.tensor(np.ascontiguousarray(x0s.T), dtype=torch.float64, device=ctx.device)   # [d][P]

    def run():
        ll = ctx.empty(P)
        X = bh.EnsemblePath(Po.tt, 2, P, ctx)
        ll2 = ctx.empty(P)
        ctx.check(ctx.lib.bhip_sample_solve(ctx.h, Po.h, None, bh.api.vp(x0_dev.data_ptr()), None, P, None, P,
                                            bh.api.vp(ll.data_ptr()), 0, P, 5, 1, 0))
        ctx.check(ctx.lib.bhip_sample_solve(ctx.h, Po.h, None, bh.api.vp(x0_dev.data_ptr()), None, P, X.ptr(), P,
                                            bh.api.vp(ll2.data_ptr()), 0, P, 5, 1, 0))
        return ll.cpu().numpy(), ll2.cpu().numpy(), X.paths()
    a, b = both(ctx, run)
    for u, v in zip(a, b):
        assert np.array_equal(u, v)
    assert np.array_equal(a[0], a[1])
    ref = case.oracle_proposal()
    for p in (0, 64, 129):
        W = o.wiener_sample(case.tt, 1, 5, p, 1)
        Xr = o.solve_guided(ref, x0s[p], W)
        assert np.array_equal(a[2][p], Xr) and a[0][p] == o.llikelihood(ref, Xr)


def test_placement_tuning_changes_nothing_but_the_allocation(ctx):
    """BHIP_OPT_TUNE_PLACEMENT: an ensemble of 1 GiB or more keeps W and Xo in two contiguous allocations and tests with two write
    streams whether they share a 96-GiB piece of the device memory (the pair, then up to seven more Xo); the state is initialised afresh
    afterwards, so chains, paths and statistics are those of an ensemble that was not placed"""
    import torch
    case = [c for c in problems.cases(1001) if c.name == "fhn_partialbridge_extreme"][0]
    Po = case.bh_proposal(bh, ctx)
    n = 36000                                            # x 32 KB of state per chain > 1 GiB
    a = bh.Chains(Po, case.x0, n, seed=9)
    info = a.placement()
    assert 1 <= info["tries"] <= 34 and info["gbs_same_piece"] > 1000 and info["gbs_kept"] > 0
    assert info["gbs_kept"] >= 0.90 * info["gbs_same_piece"]      # the pair that was kept is not slower than two streams in one piece (single runs scatter by +-10 %)
    a.step(0.9, 3)
    ctx.set_option(bh.OPT_TUNE_PLACEMENT, 0)
    try:
        b = bh.Chains(Po, case.x0, n, seed=9)
    finally:
        ctx.set_option(bh.OPT_TUNE_PLACEMENT, 1)
    assert b.placement()["tries"] == 0
    b.step(0.9, 3)
    assert np.array_equal(a.ll(), b.ll()) and np.array_equal(a.acc(), b.acc()) and torch.equal(a.stats(), b.stats())
    Xa, Wa = a.paths(n - 3, 3)
    Xb, Wb = b.paths(n - 3, 3)
    assert np.array_equal(Xa, Xb) and np.array_equal(Wa, Wb)
    small = bh.Chains(Po, case.x0, 512, seed=9)          # small ensembles are not tuned
    assert small.placement()["tries"] == 0


def test_piece_map_of_the_context_places_six_ensembles_alive_at_once():
    """The context's piece map (round 5): which of the three 96-GiB pieces the large buffers of its live ensembles lie
    in.  Six ensembles of 2.2 GB alive in ONE process: every one ends with W and Xo in different pieces (the two-stream rate of the kept
    pair clearly above the one-piece rate), the pieces recorded differ, bhip_ctx_piece_of looks a held buffer up and classifies a foreign
    one into one of the known pieces, the set-up of every ensemble after the first takes a few milliseconds and holds at most a few
    spare Xo; a destroyed ensemble leaves the map."""
    import ctypes as C
    import time
    import torch
    torch.cuda.empty_cache()                             # (blocks cached by earlier tests of this process would be handed out first)
    c2 = bh.Context(0)
    case = [c for c in problems.cases(1001) if c.name == "fhn_partialbridge_extreme"][0]
    Po = case.bh_proposal(bh, c2)
    n = 65536                                            # 1.07 GB of lines + 1.05 GB of proposal paths
    ens, setup_ms = [], []
    for k in range(6):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ch = bh.Chains(Po, case.x0, n, seed=9 + k)
        torch.cuda.synchronize()
        setup_ms.append((time.perf_counter() - t0) * 1e3)
        ens.append(ch)
    infos = [e.placement() for e in ens]
    # W and Xo apart (the mean of the kept pair's four two-stream rates >= 1.14 x the one-piece rate, PlaceParams::mean_min -- the cut the
    # timed pairs of profiles/r5_piece_map.txt put between the classes): in a process of its own 12 of 12; inside the test suite -- the
    # allocator's free lists are what two hundred earlier tests left -- one ensemble of 1-GB buffers may exhaust its candidates inside
    # one piece: at least five of six, and none below the one-piece rate
    assert sum(i["gbs_kept"] >= 1.14 * i["gbs_same_piece"] for i in infos) >= 5, (infos, setup_ms)
    for k, i in enumerate(infos):
        assert 1 <= i["tries"] <= 34, (k, infos)                    # Xo candidates (24 + 4 + 4 at most) + further W runs (2)
        assert i["gbs_kept"] >= 0.90 * i["gbs_same_piece"], (k, infos, setup_ms)
        # the labels: ids of the context's map, or -1 for a buffer the tests could not attribute (they are bookkeeping: what decides is the pair test)
        assert i["piece_w"] in (-1, 0, 1, 2) and i["piece_xo"] in (-1, 0, 1, 2), (k, infos)
    assert len({i["gbs_same_piece"] for i in infos}) == 1                   # the reference rate is the context's, measured once
    assert infos[0]["piece_w"] == 0 and infos[0]["piece_xo"] == 1           # the first ensemble founds the map
    assert len(({i["piece_w"] for i in infos} | {i["piece_xo"] for i in infos}) - {-1}) >= 2
    # look-up of a held buffer, classification of a foreign one (its contents are overwritten: a scratch tensor)
    Xo_dev, ld = C.c_void_p(), C.c_long()
    c2.check(c2.lib.bhip_chains_proposal_X(ens[0].h, C.byref(Xo_dev), C.byref(ld)))
    pc = C.c_int(-7)
    c2.check(c2.lib.bhip_ctx_piece_of(c2.h, Xo_dev, C.c_size_t(8 * 1001 * 2 * ld.value), C.byref(pc)))
    assert pc.value == infos[0]["piece_xo"] == 1
    scratch = torch.empty(1 << 27, dtype=torch.float64, device=c2.device)            # 1 GiB
    c2.check(c2.lib.bhip_ctx_piece_of(c2.h, C.c_void_p(scratch.data_ptr()), C.c_size_t(scratch.numel() * 8), C.byref(pc)))
    assert pc.value in (-1, 0, 1, 2)
    # the ensembles still run, and give what an unplaced ensemble gives -- the LAST one, and the FIRST one too: its W and Xo were the
    # map's representatives while five later ensembles and a foreign buffer were classified with write streams against them
    # (advisor r5: the tested ranges of a representative are saved and restored; before that the first ensemble's chain state was overwritten)
    ens[5].step(0.9, 2)
    ens[0].step(0.9, 2)
    X0, W0 = ens[0].paths(0, 4)
    XL, WL = ens[0].paths(n - 4, 4)                      # head and tail of the buffers: where the streams went
    c2.set_option(bh.OPT_TUNE_PLACEMENT, 0)
    ref = bh.Chains(Po, case.x0, n, seed=14)
    ref.step(0.9, 2)
    assert np.array_equal(ens[5].ll(), ref.ll()) and np.array_equal(ens[5].acc(), ref.acc())
    del ref
    ref0 = bh.Chains(Po, case.x0, n, seed=9)
    ref0.step(0.9, 2)
    assert np.array_equal(ens[0].ll(), ref0.ll()) and np.array_equal(ens[0].acc(), ref0.acc())
    for got, want in ((X0, ref0.paths(0, 4)[0]), (W0, ref0.paths(0, 4)[1]), (XL, ref0.paths(n - 4, 4)[0]), (WL, ref0.paths(n - 4, 4)[1])):
        assert np.array_equal(got, want)
    ref = ref0
    # every set-up after the first: W sample + solve of 65 536 chains plus the tests -- tens of milliseconds at most, not round 4's 23-60 on top
    assert sorted(setup_ms[1:])[2] < 150.0, (setup_ms, infos)      # (the median; an ensemble that went through all its candidates takes longer)
    del ens, ref
    torch.cuda.empty_cache()

