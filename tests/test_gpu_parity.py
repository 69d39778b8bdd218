"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI via the
host mirror, against the CPU oracle on identical inputs.

Tolerances (fp64):
  * models whose drift is polynomial (OU, LinPro, FitzHugh-Nagumo, Lorenz): paths X, Wiener paths W,
    log-likelihoods ll, accept decisions and acceptance counts are BIT-EXACT (compared with ==, i.e.
    up to the sign of zero).  Both sides round every operation identically: same operation order,
    no FMA contraction, IEEE division / sqrt, and a libm-free log / sincos in the RNG.
  * models calling sin() (NCLAR, IntegratedDiffusion, Pendulum): device sin (ocml) and host sin
    (glibc) may differ by 1 ulp per call; the guide's stiffness near T amplifies that.  Stated
    tolerance: |X - X_oracle| <= 1e-9 * (1 + max|X|), |ll - ll_oracle| <= 1e-8 * (1 + |ll|).
"""
import math

import numpy as np
import pytest
import torch

import bridgehip as bh
import oracle as o
import problems

pytestmark = pytest.mark.gpu

SEED = 20240928


@pytest.fixture(scope="module")
def ctx():
    c = bh.default_context(0)
    # the product must be the native library, not a fallback
    assert bh._lib.SO_PATH.endswith("libbridgehip.so") and bh._lib.load().bhip_device_count() >= 1
    return c


def check_paths(case, X_gpu, X_ref, what="X", libm_trig=False):
    # libm_trig: the reference values were frozen before the shared fdlibm-form sin / cos (goldens v1, v2)
    if case.exact and not (libm_trig and case.trig):
        assert np.array_equal(X_gpu, X_ref), f"{case.name}: {what} not bit-exact, max diff {np.abs(X_gpu - X_ref).max():.3e}"
    else:
        tol = 1e-9 * (1 + np.abs(X_ref).max())
        assert np.abs(X_gpu - X_ref).max() <= tol, f"{case.name}: {what} diff {np.abs(X_gpu - X_ref).max():.3e} > {tol:.1e}"


def check_ll(case, ll_gpu, ll_ref, libm_trig=False):
    if case.exact and not (libm_trig and case.trig):
        assert np.array_equal(ll_gpu, ll_ref), f"{case.name}: ll not bit-exact, max diff {np.abs(ll_gpu - ll_ref).max():.3e}"
    else:
        assert np.all(np.abs(ll_gpu - ll_ref) <= 1e-8 * (1 + np.abs(ll_ref))), f"{case.name}: ll diff {np.abs(ll_gpu - ll_ref).max():.3e}"


# --------------------------------------------------------------------------- LOOP A
@pytest.mark.parametrize("mp", [1, 2, 3, 5, 6, 8, 12, 32])     # (<= 4: registers; multiples of 4 above: blocks of four components; else the generic kernel)
def test_wiener_sample_bit_exact(ctx, mp):
    tt = problems.tau_grid(2.0, 257)
    P = 70                      # not a multiple of the wave size: exercises the tail
    W = bh.sample(tt, bh.Wiener(mp), npaths=P, seed=SEED, iter=3, path0=1000, ctx=ctx)
    Wh = W.paths()
    for p in (0, 1, 63, 64, 69):
        assert np.array_equal(Wh[p], o.wiener_sample(tt, mp, SEED, 1000 + p, 3))
    assert np.all(Wh[:, 0, :] == 0.0)


def test_device_normals_match_host_spec(ctx):
    # W on a grid with dt = 1 is the cumulated sum of the raw normals: checks Philox + Box-Muller on device
    tt = np.arange(0, 1025, dtype=np.float64)
    W = bh.sample(tt, bh.Wiener(1), npaths=3, seed=5, iter=0, path0=2 ** 32 - 3, ctx=ctx).paths()       # the largest path ids
    for p in range(3):
        z = o.normals(5, 2 ** 32 - 3 + p, 0, 0, 1024)
        assert np.array_equal(W[p, :, 0], np.concatenate([[0.0], np.cumsum(z)]))


# --------------------------------------------------------------------------- LOOP B, unguided
@pytest.mark.parametrize("case", problems.forward_cases(301), ids=lambda c: c.name)
def test_forward_euler_maruyama(ctx, case):
    P = 70
    Wh = np.stack([o.wiener_sample(case.tt, case.mp, SEED, p, 0) for p in range(P)])
    W = bh.EnsemblePath.from_paths(case.tt, Wh, ctx)
    proc = case.bh_proposal(bh, ctx)
    X = bh.solve(bh.EulerMaruyama(), case.x0, W, proc)
    Xh = X.paths()
    ref = np.stack([o.solve_em(case.model, case.d, case.mp, case.par, case.tt, case.x0, Wh[p]) for p in range(P)])
    check_paths(case, Xh, ref)
    assert np.array_equal(Xh[:, 0, :], np.tile(case.x0, (P, 1)))      # yy[1] = u stored before the update


def test_K1_manual_doctest_vector_on_device(ctx):
    import json
    import os
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "manual_ou.json")))
    W = bh.EnsemblePath.from_paths(g["tt"], np.array(g["W"])[None, :, None], ctx)
    proc = bh.PlainProcess(g["tt"], bh.OrnsteinUhlenbeck(g["beta"], g["sigma"]), ctx=ctx)
    X = bh.solve(bh.Euler(), g["x0"], W, proc).paths()[0, :, 0]
    assert np.abs(X - np.array(g["X"])).max() < 5e-6


# --------------------------------------------------------------------------- LOOP B + C, guided, external W
@pytest.mark.parametrize("case", problems.cases(301), ids=lambda c: c.name)
def test_guided_solve_and_llikelihood(ctx, case):
    P = 70
    Po_ref = case.oracle_proposal()
    Po = case.bh_proposal(bh, ctx)
    Wh = np.stack([o.wiener_sample(case.tt, case.mp, SEED, p, 0) for p in range(P)])
    W = bh.EnsemblePath.from_paths(case.tt, Wh, ctx)
    Y = bh.EnsemblePath(np.zeros_like(case.tt), case.d, P, ctx)
    ll = ctx.empty(P)
    end = bh.solve_(bh.Euler(), Y, case.x0, W, Po, ll=ll)
    assert np.array_equal(Y.tt, case.tt)                               # tt[:] = P.tt  src/euler.jl:256
    Xh = Y.paths()
    Xref = np.stack([o.solve_guided(Po_ref, case.x0, Wh[p]) for p in range(P)])
    llref = np.array([o.llikelihood(Po_ref, Xref[p]) for p in range(P)])
    assert np.all(np.isfinite(Xref)) and np.all(np.isfinite(llref))
    check_paths(case, Xh, Xref)
    check_ll(case, ll.cpu().numpy(), llref)
    assert np.array_equal(end.cpu().numpy().T, Xh[:, -1, :])           # solve! returns yy[N]  :267
    # stand-alone llikelihood on the stored ensemble == fused value, also with skip
    ll2 = bh.llikelihood(bh.LeftRule(), Y, Po)
    assert torch.equal(ll2, ll)
    ll3 = bh.llikelihood(bh.LeftRule(), Y, Po, skip=7).cpu().numpy()
    ref3 = np.array([o.llikelihood(Po_ref, Xh[p], skip=7) for p in range(P)])
    check_ll(case, ll3, ref3) if case.exact else None
    # bridge! alias
    Y2 = Y.copy()
    bh.bridge_(Y2, case.x0, W, Po)
    assert torch.equal(Y2.data, Y.data)


def test_guided_endpoint_rule(ctx):
    c = [k for k in problems.cases(301) if k.name == "ou_guidedbridge"][0]
    Po = c.bh_proposal(bh, ctx)
    X, _, _ = bh.sample_solve(c.x0, Po, 256, seed=1)
    assert torch.all(X.data[-1, 0] == c.v[0])                          # Hd[N] = 0  =>  X[N] = V[N] = v
    c = [k for k in problems.cases(301) if k.name == "ou_guidedbridge_free_end"][0]
    Po = c.bh_proposal(bh, ctx)
    X, _, _ = bh.sample_solve(c.x0, Po, 256, seed=1)
    assert X.data[-1, 0].std() > 1e-3                                   # hT != 0: free endpoint


def test_per_path_starting_points_chain_segments(ctx):
    # solve! returns the endpoint so that segments can be chained (test/smoothing.jl:88-92)
    c = [k for k in problems.cases(201) if k.name == "linpro2_guidedbridge"][0]
    Po = c.bh_proposal(bh, ctx)
    P = 130
    X1, W1, _ = bh.sample_solve(c.x0, Po, P, seed=3, iter=0, store_W=True)
    start = X1.data[100].contiguous()                                   # [d, P] states at grid index 100
    tt2 = c.tt[100:].copy()
    c2 = problems.Case("seg2", tt2, c.x0, c.model, c.par, c.aux, c.apar, c.kind, c.d, c.mp, v=c.v)
    Po2 = c2.bh_proposal(bh, ctx)
    W2 = bh.EnsemblePath(tt2, c.mp, P, ctx, W1.data[100:].contiguous())
    X2 = bh.solve(bh.Euler(), start, W2, Po2)
    ref = c2.oracle_proposal()
    Xh, Wh, sh = X2.paths(), W2.paths(), start.cpu().numpy()
    for p in (0, 64, 129):
        assert np.array_equal(Xh[p], o.solve_guided(ref, sh[:, p], Wh[p]))


# --------------------------------------------------------------------------- fused A+B+C
@pytest.mark.parametrize("case", [c for c in problems.cases(301) if c.name in
                                  ("ou_guidedbridge", "fhn_partialbridge_first", "fhn_nuh", "fhn_inplace", "nclar_firstcomponent",
                                   "linpro3_guidedbridge", "fhn2_nuh_full", "linpro3_partial_m2")], ids=lambda c: c.name)
def test_fused_sample_solve_equals_separate_passes(ctx, case):
    P = 200
    Po = case.bh_proposal(bh, ctx)
    X, W, ll = bh.sample_solve(case.x0, Po, P, seed=SEED, iter=2, path0=17, store_W=True)
    W2 = bh.sample(case.tt, bh.Wiener(case.mp), npaths=P, seed=SEED, iter=2, path0=17, ctx=ctx)
    assert torch.equal(W.data, W2.data)
    ll2 = ctx.empty(P)
    X2 = bh.solve(bh.Euler(), case.x0, W2, Po, ll=ll2)
    assert torch.equal(X.data, X2.data) and torch.equal(ll, ll2)
    # and against the oracle's four-pass ensemble driver
    _, llref, last = o.ensemble_proposals(case.oracle_proposal(), case.x0, P, 17, SEED, 2, want_last=True)
    check_ll(case, ll.cpu().numpy(), llref)
    check_paths(case, X.data[-1].cpu().numpy().T, last, "X[N]")
    # ll-only mode (no path stored) gives the same weights
    _, _, ll3 = bh.sample_solve(case.x0, Po, P, seed=SEED, iter=2, path0=17, store_X=False)
    assert torch.equal(ll3, ll)


def test_results_do_not_depend_on_launch_geometry_or_sharding(ctx):
    # the multi-GPU run shards global path ids; any split must give identical paths
    c = [k for k in problems.cases(301) if k.name == "fhn_partialbridge_extreme"][0]
    Po = c.bh_proposal(bh, ctx)
    X, _, ll = bh.sample_solve(c.x0, Po, 1000, seed=9, iter=1, path0=0)
    Xa, _, lla = bh.sample_solve(c.x0, Po, 333, seed=9, iter=1, path0=0)
    Xb, _, llb = bh.sample_solve(c.x0, Po, 667, seed=9, iter=1, path0=333)
    assert torch.equal(X.data[:, :, :333], Xa.data) and torch.equal(X.data[:, :, 333:], Xb.data)
    assert torch.equal(ll, torch.cat([lla, llb]))


# --------------------------------------------------------------------------- pCN MCMC
@pytest.mark.parametrize("case", [c for c in problems.cases(151) if c.name in
                                  ("fhn_partialbridge_first", "fhn_partialbridge_extreme", "fhn_startend", "ou_guidedbridge",
                                   "fhn_inplace", "nclar_firstcomponent", "intdiff_partialbridge", "linpro3_partial_m2",
                                   "linpro2_guidedbridge", "fhn2_nuh_full")],        # noise dimension 1 and 2: line layout; 3: slots
                         ids=lambda c: c.name)
def test_mcmc_chains_match_oracle(ctx, case):
    nch, iters = 70, 25
    ch = bh.Chains(case.bh_proposal(bh, ctx), case.x0, nch, seed=SEED, path0=5)
    ll0 = ch.ll()
    ch.step(case.rho, iters)
    X, W = ch.paths()
    ll, acc = ch.ll(), ch.acc()
    Po_ref = case.oracle_proposal()
    for p in (0, 1, 63, 64, 69):
        r = o.mcmc(Po_ref, case.x0, case.rho, iters, SEED, 5 + p)
        if case.exact:
            assert acc[p] == r["acc"] and ll[p] == r["ll"]
            assert np.array_equal(W[p], r["W"]) and np.array_equal(X[p], r["X"])
        else:
            # an accept decision can flip when llo - ll is within rounding of log(U): compare only if it did not
            if acc[p] == r["acc"]:
                assert abs(ll[p] - r["ll"]) <= 1e-8 * (1 + abs(r["ll"]))
                assert np.abs(W[p] - r["W"]).max() <= 1e-12 and np.abs(X[p] - r["X"]).max() <= 1e-9 * (1 + np.abs(r["X"]).max())
    assert 0 <= acc.min() and acc.max() <= iters
    # the device stats block
    st = ch.stats().cpu().numpy()
    assert st[0] == nch and st[1] == iters and st[2] == acc.sum() and st[7] == (acc.astype(float) ** 2).sum()
    assert abs(st[3] - ll.sum()) <= 1e-10 * np.abs(ll).sum() and st[5] == ll.min() and st[6] == ll.max()
    assert abs(st[4] - (ll ** 2).sum()) <= 1e-10 * (ll ** 2).sum()
    # pointwise ensemble statistics == numpy over the downloaded current paths
    n, mean, m2 = ch.pathstats()
    assert n == nch and np.allclose(mean, X.mean(0), rtol=1e-12, atol=1e-13)
    dev = X - X.mean(0)
    assert np.allclose(m2, np.einsum("pir,pic->irc", dev, dev), rtol=1e-9, atol=1e-12)
    assert np.all(np.isfinite(ll0))


def test_mcmc_driver_and_subsamples(ctx):
    c = [k for k in problems.cases(101) if k.name == "fhn_partialbridge_extreme"][0]
    out = bh.mcmc(c.bh_proposal(bh, ctx), c.x0, 30, 0.9, nchains=128, seed=4, subsamples=range(0, 31, 10))
    assert len(out["XX"]) == 4 and out["XX"][0].shape == (128, 101, 2)
    assert 0 < out["acc"].sum() < 30 * 128                               # test/partialbridge.jl:119  1 < acc < iterations
    r = o.mcmc(c.oracle_proposal(), c.x0, 0.9, 30, 4, 77)
    assert out["acc"][77] == r["acc"] and np.array_equal(out["XX"][-1][77], r["X"])
    # chains without X storage carry the same (W, ll) state
    ch = bh.Chains(c.bh_proposal(bh, ctx), c.x0, 128, seed=4, store_X=False)
    ch.step(0.9, 30)
    assert np.array_equal(ch.ll(), out["ll"]) and np.array_equal(ch.acc(), out["acc"])
    # the current X is a function of the current W: available even when no Xo was ever stored
    assert np.array_equal(ch.paths(want_W=False)[0], out["XX"][-1])
    with pytest.raises(bh.BridgeError):
        ch.proposal_X()
    # with store_X the proposal buffer holds Xo of the last iteration; it equals the current X
    # exactly for the chains that accepted it
    ch2 = out["chains"]
    acc_before = ch2.acc()
    ch2.step(0.9, 1)
    accepted = (ch2.acc() - acc_before).astype(bool)
    Xo = ch2.proposal_X().cpu().numpy()          # [N, d, n]
    Xc = ch2.current_X().data.cpu().numpy()
    assert accepted.any() and (~accepted).any()
    assert np.array_equal(Xo[:, :, accepted], Xc[:, :, accepted])
    assert not np.array_equal(Xo[:, :, ~accepted], Xc[:, :, ~accepted])


# --------------------------------------------------------------------------- reference-style error behaviour
def test_reference_error_messages(ctx):
    c = problems.cases(51)[3]
    Po = c.bh_proposal(bh, ctx)
    W = bh.sample(c.tt, bh.Wiener(1), npaths=8, ctx=ctx)
    Yshort = bh.EnsemblePath(c.tt[:-1], 2, 8, ctx)
    with pytest.raises(bh.BridgeError, match="Y and W differ in length."):
        bh.solve_(bh.Euler(), Yshort, c.x0, W, Po)
    with pytest.raises(bh.BridgeError, match="Starting point has wrong length."):
        bh.solve(bh.Euler(), [0.0, 0.0, 0.0], W, Po)
    plain = bh.PlainProcess(c.tt, c.bh_process(bh), ctx=ctx)
    with pytest.raises(bh.BridgeError):
        bh.llikelihood(bh.LeftRule(), bh.solve(bh.Euler(), c.x0, W, plain), plain)


def test_inline_philox_equals_rocrand_device_generator(tmp_path):
    """the noise specification is rocRAND's default PHILOX4_32_10 with an explicit counter layout:
    compile tests/rocrand_check.hip against rocRAND's device API and compare word for word"""
    import os
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available on this box")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "rocrand_check")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O2", "-I", os.path.join(root, "bridge.jl_amd", "csrc"),
                           os.path.join(root, "tests", "rocrand_check.hip"), "-o", exe], stderr=subprocess.DEVNULL)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.startswith("OK"), out.stdout + out.stderr


def test_device_forms_of_the_generator_are_bit_identical(tmp_path):
    """bhip_rng.h evaluates the uniforms and the square root of Box-Muller with shorter device sequences;
    tests/rng_device_forms.hip compares them with the portable expressions (integer->double conversion, IEEE sqrt)
    on 2^32 inputs from the ranges used, and the LDS copy of the tables with the constant-memory one"""
    import os
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available on this box")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "rng_device_forms")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O2", "-ffp-contract=off", "-I", os.path.join(root, "bridge.jl_amd", "csrc"),
                           os.path.join(root, "tests", "rng_device_forms.hip"), "-o", exe], stderr=subprocess.DEVNULL)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.startswith("OK"), out.stdout + out.stderr


@pytest.mark.parametrize("case", problems.cases(101) + problems.forward_cases(101), ids=lambda c: c.name)
def test_committed_golden_vectors_on_device(ctx, case):
    """the frozen vectors (tests/golden/guided_paths_v5.npz, noise specification v4) through the C ABI: in-kernel noise, guided
    solve, fused log-likelihood and a pCN chain reproduce them bit for bit without the oracle being involved"""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "guided_paths_v5.npz"))
    N, npaths, seed, iters = (int(v) for v in g["meta"])
    rho = float(g["rho"])
    Po = case.bh_proposal(bh, ctx)
    X, W, ll = bh.sample_solve(case.x0, Po, npaths, seed=seed, store_W=True)
    assert np.array_equal(W.paths(), g[case.name + "/W"])
    check_paths(case, X.paths(), g[case.name + "/X"])
    if case.kind == o.GUIDE_NONE:
        return
    check_ll(case, ll.cpu().numpy(), g[case.name + "/ll"])
    if case.exact and (case.name + "/chain_W") in g.files:
        ch = bh.Chains(Po, case.x0, 2, seed=seed)
        ch.step(rho, iters)
        Xc, Wc = ch.paths(1, 1)
        assert np.array_equal(Wc[0], g[case.name + "/chain_W"]) and np.array_equal(Xc[0], g[case.name + "/chain_X"])
        assert ch.ll()[1] == g[case.name + "/chain_ll_acc"][0] and ch.acc()[1] == g[case.name + "/chain_ll_acc"][1]


@pytest.mark.parametrize("version", ["v1", "v2", "v3", "v4"])
@pytest.mark.parametrize("case", problems.cases(101) + problems.forward_cases(101), ids=lambda c: c.name)
def test_earlier_golden_paths_given_their_wiener_paths_on_device(ctx, case, version):
    """guided_paths_v1.npz (noise specification v1), _v2.npz, _v3.npz (specification v2), _v4.npz (specification v3): the guided paths and
    log-likelihoods GIVEN their stored Wiener paths do not involve the generator; the external-W solve must still reproduce
    them (the frozen arithmetic of rounds 1 and 2 guards the solver across the changes of the noise specification)"""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", f"guided_paths_{version}.npz"))
    lt = version in ("v1", "v2")          # written before the shared fdlibm-form sin / cos
    Po = case.bh_proposal(bh, ctx)
    W = bh.EnsemblePath.from_paths(case.tt, g[case.name + "/W"], ctx)
    if case.kind == o.GUIDE_NONE:
        X = bh.solve(bh.EulerMaruyama(), case.x0, W, Po)
        check_paths(case, X.paths(), g[case.name + "/X"], libm_trig=lt)
        return
    ll = ctx.empty(W.npaths)
    X = bh.solve(bh.Euler(), case.x0, W, Po, ll=ll)
    check_paths(case, X.paths(), g[case.name + "/X"], libm_trig=lt)
    check_ll(case, ll.cpu().numpy(), g[case.name + "/ll"], libm_trig=lt)


def test_chains_with_skip_match_oracle(ctx):
    """llikelihood(...; skip): the ensemble's skip applies to the initial ll AND (by default) to every llo, as in
    partialbridge_nclar.jl:121; compared with the oracle chain run with the same skip"""
    c = [k for k in problems.cases(161) if k.name == "fhn_partialbridge_extreme"][0]
    Po, ref = c.bh_proposal(bh, ctx), c.oracle_proposal()
    for skip in (1, 7):
        ch = bh.Chains(Po, c.x0, 130, seed=5, skip=skip)
        ch.step(0.9, 6)                       # default: the ensemble's skip
        out = bh.mcmc(Po, c.x0, 6, 0.9, nchains=130, seed=5, skip=skip)
        assert np.array_equal(out["ll"], ch.ll()) and np.array_equal(out["acc"], ch.acc())
        for p in (0, 129):
            r = o.mcmc(ref, c.x0, 0.9, 6, 5, p, skip=skip)
            assert ch.ll()[p] == r["ll"] and ch.acc()[p] == r["acc"]
        lib = ctx.lib
        ch2 = bh.Chains(Po, c.x0, 130, seed=5, skip=skip)
        ctx.check(lib.bhip_chains_step(ch2.h, 0.9, 6, -1))     # BHIP_SKIP_OF_INIT through the C ABI
        assert np.array_equal(ch2.ll(), ch.ll())


def test_batched_chain_steps_equal_single_steps(ctx):
    """bhip_chains_step(iters = k) skips the proposal-path store of all but its last iteration (the buffer would be
    overwritten unseen): the chain state, the statistics and the final Xo must equal k calls with iters = 1"""
    c = [k for k in problems.cases(201) if k.name == "fhn_partialbridge_extreme"][0]
    Po = c.bh_proposal(bh, ctx)
    a, b = bh.Chains(Po, c.x0, 300, seed=12), bh.Chains(Po, c.x0, 300, seed=12)
    a.step(0.9, 7)
    for _ in range(7):
        b.step(0.9, 1)
    assert np.array_equal(a.ll(), b.ll()) and np.array_equal(a.acc(), b.acc())
    assert torch.equal(a.proposal_X(), b.proposal_X()) and torch.equal(a.current_X().data, b.current_X().data)
    assert torch.equal(a.stats(), b.stats())


@pytest.mark.parametrize("name", ["fhn_partialbridge_extreme", "fhn2_nuh_full", "linpro3_partial_m2"])
def test_chain_checkpoint_and_resume(ctx, name):
    """Chains.save() / load(): (current W, ll, acceptance counts, iteration counter) + the counter-based noise make a
    resumed ensemble continue with exactly the iterations the original runs (line layout, m' = 1 and 2; slots, m' = 3)"""
    c = [k for k in problems.cases(121) if k.name == name][0]
    Po = c.bh_proposal(bh, ctx)
    a = bh.Chains(Po, c.x0, 200, seed=21, path0=9)
    a.step(0.9, 7)
    state = a.save()
    a.step(0.9, 6)
    b = bh.Chains(Po, c.x0, 200, seed=21, path0=9)          # a fresh ensemble (iteration 0) ...
    b.load(state)                                            # ... put into the saved state
    assert b.iterations == 7
    b.step(0.9, 6)
    Xa, Wa = a.paths(0, 200)
    Xb, Wb = b.paths(0, 200)
    assert np.array_equal(a.ll(), b.ll()) and np.array_equal(a.acc(), b.acc())
    assert np.array_equal(Wa, Wb) and np.array_equal(Xa, Xb) and torch.equal(a.stats(), b.stats())
    # a state only fits the ensemble it came from
    with pytest.raises(bh.BridgeError, match="seed"):
        bh.Chains(Po, c.x0, 200, seed=22, path0=9).load(state)
    with pytest.raises(bh.BridgeError, match="shape"):
        bh.Chains(Po, c.x0, 100, seed=21, path0=9).load(state)
    with pytest.raises(bh.BridgeError, match="not a chain state"):
        b.load(np.zeros(len(state), dtype=np.uint8))
    # ... and the noise specification it was driven with (advisor r2: a v1 state, whose header field was 0, must not resume silently)
    old = state.copy()
    old[60:64] = 0                                           # ChainStateHeader.rng_spec
    with pytest.raises(bh.BridgeError, match="noise specification"):
        b.load(old)
    assert int(np.frombuffer(state[60:64].tobytes(), dtype=np.int32)[0]) == 4   # bhip-philox-v4
    old[60:64] = np.frombuffer(np.int32(2).tobytes(), dtype=np.uint8)           # a round-2 (v2) state
    with pytest.raises(bh.BridgeError, match="bhip-philox-v2"):
        b.load(old)
