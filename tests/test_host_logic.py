"""CPU tests of the product's host side (no GPU needed): the C-ABI library loads and exports every
symbol include/bridgehip.h declares, the host-side guide pre-computation (C++ in libbridgehip.so,
run through a device=-1 host-only context) agrees BIT FOR BIT with the CPU oracle, the RNG spec
matches, and errors are reported the way the header promises.  No compute kernels are launched.
"""
import ctypes as C
import os
import re

import numpy as np
import pytest

import bridgehip as bh
import oracle as o
import problems

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hctx():
    return bh.Context(-1)


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "bridgehip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(bhip_\w+)\s*\(", hdr)) - {"bhip_aux_fn"}
    assert len(declared) >= 35
    lib = C.CDLL(bh._lib.SO_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in bridgehip.h but not exported"
    assert declared == set(bh._lib.SIGNATURES), declared ^ set(bh._lib.SIGNATURES)
    assert bh._lib.load().bhip_version() == 200


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(bh.BridgeError):
        bh.Context(0)
    h = C.c_void_p()
    assert bh._lib.load().bhip_ctx_create(0, None, C.byref(h)) != 0


def test_host_only_context_refuses_device_work(hctx):
    c = problems.cases(21)[3]
    Po = c.bh_proposal(bh, ctx=hctx)
    lib = hctx.lib
    dev = C.c_void_p()
    assert lib.bhip_malloc(hctx.h, 64, C.byref(dev)) == -2
    assert b"host-only" in lib.bhip_last_error(hctx.h)
    ch = C.c_void_p()
    assert lib.bhip_chains_create(hctx.h, Po.h, 4, 0, 1, 1, C.byref(ch)) == -2
    z = np.zeros(2)
    assert lib.bhip_sample_solve(hctx.h, Po.h, bh.api._dptr(z), None, None, 4, None, 4, None, 0, 4, 1, 0, 0) == -2


def test_rng_spec_matches_oracle_bitwise():
    lib = bh._lib.load()
    for ctr, key in (([0, 0, 0, 0], [0, 0]), ([1, 2, 3, 4], [5, 6]), ([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2)):
        c, k, out = (C.c_uint32 * 4)(*ctr), (C.c_uint32 * 2)(*key), (C.c_uint32 * 4)()
        lib.bhip_philox4x32_10(c, k, out)
        assert list(out) == o.philox(ctr, key)
    for seed, path, it, n0, n in ((1, 0, 0, 0, 1001), (0xDEADBEEFCAFE, 77, 3, 5, 300), (2 ** 63 + 5, 2 ** 32 - 1, 2 ** 31, 1, 64)):
        z = np.empty(n)
        lib.bhip_normals_host(seed, path, it, n0, n, bh.api._dptr(z))
        assert np.array_equal(z, o.normals(seed, path, it, n0, n))
        for spec in (4, 3, 2):    # BHIP_OPT_NOISE_SPEC: the product's host form of every stream against the oracle's
            zs = np.empty(n)
            lib.bhip_normals_host_spec(spec, seed, path, it, n0, n, bh.api._dptr(zs))
            with o.noise_spec(spec):
                assert np.array_equal(zs, o.normals(seed, path, it, n0, n)), spec
            assert np.array_equal(zs, z) == (spec == 4)


@pytest.mark.parametrize("case", problems.cases(101), ids=lambda c: c.name)
def test_host_guide_equals_oracle_bitwise(hctx, case):
    g = case.oracle_guide()
    Po = case.bh_proposal(bh, ctx=hctx)
    if case.kind == o.GUIDE_HV:
        assert np.array_equal(Po.Hd, g["Hd"]) and np.array_equal(Po.V, g["V"])
    elif case.kind == o.GUIDE_LMMU:
        assert np.array_equal(Po.L, g["L"]) and np.array_equal(Po.mu, g["mu"])
        assert np.array_equal(Po.M, g["M"])
    else:
        assert np.array_equal(Po.nu, g["nu"]) and np.array_equal(Po.H, g["H"])
        if case.kind == o.GUIDE_NUH:
            assert Po.C == g["C"]
    N, d, mp, m, kind = (C.c_int() for _ in range(5))
    hctx.check(hctx.lib.bhip_proposal_info(Po.h, *map(C.byref, (N, d, mp, m, kind))))
    assert (N.value, d.value, mp.value, kind.value) == (len(case.tt), case.d, case.mp, case.kind)


@pytest.mark.parametrize("kind", [o.GUIDE_HV, o.GUIDE_NUH], ids=["guidedbridge", "nuh"])
def test_host_guide_d32_equals_oracle_bitwise(hctx, kind):
    """config C5: the dimension-generic host side (LU inverse / solve for n > 3) against the oracle"""
    c = problems.linpro_big_case(32, 101, kind)
    g = c.oracle_guide()
    Po = c.bh_proposal(bh, ctx=hctx)
    if kind == o.GUIDE_HV:
        assert np.array_equal(Po.Hd, g["Hd"]) and np.array_equal(Po.V, g["V"])
        # closed form: Hdiamond(t) = phim*lam*phim' - lam for the auxiliary LinPro (src/linpro.jl:115-126)
        from scipy.linalg import expm, solve_continuous_lyapunov
        d = 32
        B = o.uncm(c.apar[:d * d], d, d)
        sig = o.uncm(c.apar[d * d + d:], d, d)
        lam = solve_continuous_lyapunov(B, -sig @ sig.T)
        phim = expm(-(1.0 - c.tt[50]) * B)
        assert np.abs(Po.Hd[50] - (phim @ lam @ phim.T - lam)).max() < 1e-6
    else:
        assert np.array_equal(Po.nu, g["nu"]) and np.array_equal(Po.H, g["H"]) and Po.C == g["C"]


def test_lptilde_matches_oracle(hctx):
    c = problems.cases(101)[0]
    Po = c.bh_proposal(bh, ctx=hctx)
    g = c.oracle_guide()
    ref = o.logpdfnormal(g["V"][0] - c.x0, g["Hd"][0]) - o.traceB(c.tt, 1, c.aux, c.apar)
    assert bh.lptilde(Po, c.x0) == ref
    c = [k for k in problems.cases(101) if k.name == "fhn_nuh"][0]
    Po = c.bh_proposal(bh, ctx=hctx)
    g = c.oracle_guide()
    w = g["nu"][0] - c.x0
    assert abs(bh.lptilde(Po, c.x0) - (-0.5 * w @ g["H"][0] @ w - g["C"])) < 1e-9 * abs(g["C"])


def test_user_supplied_guide_arrays_roundtrip(hctx):
    c = [k for k in problems.cases(51) if k.name == "fhn_partialbridge_first"][0]
    g = c.oracle_guide()
    P, Pt = c.bh_process(bh), c.bh_aux(bh)
    Po = bh.ProposalFromArrays(c.tt, P, Pt, bh.GUIDE_LMMU, 1, o.cm(g["L"]), o.cm(g["M"]), g["mu"], np.array(c.v), ctx=hctx)
    L2, M2, mu2 = np.empty((51, 2)), np.empty((51, 1)), np.empty((51, 1))
    hctx.check(hctx.lib.bhip_proposal_guide_get(Po.h, bh.api._dptr(L2), bh.api._dptr(M2), bh.api._dptr(mu2), None))
    assert np.array_equal(L2, o.cm(g["L"])) and np.array_equal(M2[:, 0], g["M"][:, 0, 0]) and np.array_equal(mu2, g["mu"])


def test_callback_auxiliary_reproduces_builtin(hctx):
    """a user-defined (Julia @cfunction-style) auxiliary must give the built-in's guide"""
    c = [k for k in problems.cases(51) if k.name == "fhn_startend"][0]
    p = c.apar

    def aux(t):
        lam = (t - p[5]) / (p[7] - p[5])
        uv = p[8] * lam + p[6] * (1 - lam)
        B = [[1 / p[0] - 3 * (uv * uv) / p[0], -1 / p[0]], [p[2], -1.0]]
        beta = [p[1] / p[0] + 2 * (uv * uv * uv) / p[0], p[3]]
        return B, beta, [[0.0, 0.0], [0.0, p[4] * p[4]]]

    P = c.bh_process(bh)
    ref = c.bh_proposal(bh, ctx=hctx)
    Po = bh.PartialBridge(c.tt, P, bh.CallbackAux(2, aux), c.L, c.v, c.Sigma, ctx=hctx)
    assert np.array_equal(Po.L, ref.L) and np.array_equal(Po.mu, ref.mu)
    assert np.allclose(Po.M, ref.M, rtol=1e-12)      # -(L a)L' instead of -outer(L sigma): same value, other rounding
    Po2 = bh.PartialBridgeNuH(c.tt, P, bh.CallbackAux(2, aux), c.L, c.v, 1e-3, c.Sigma, ctx=hctx)
    ref2 = bh.PartialBridgeNuH(c.tt, P, c.bh_aux(bh), c.L, c.v, 1e-3, c.Sigma, ctx=hctx)
    assert np.array_equal(Po2.nu, ref2.nu) and np.array_equal(Po2.H, ref2.H)


def test_error_reporting(hctx):
    lib, h = hctx.lib, hctx.h
    tt = np.linspace(0, 1, 11)
    out = C.c_void_p()
    par = np.array([0.1, 0.0, 1.5, 0.8, 0.3])
    dp = bh.api._dptr
    assert lib.bhip_proposal_create(h, dp(tt), 11, 99, 2, dp(par), 5, C.byref(out)) == -1
    assert b"unknown model" in lib.bhip_last_error(h)
    assert lib.bhip_proposal_create(h, dp(tt), 11, bh.MODEL_FHN, 2, dp(par), 4, C.byref(out)) == -1
    assert lib.bhip_proposal_create(h, dp(tt), 11, bh.MODEL_FHN, 3, dp(par), 5, C.byref(out)) == -1
    bad = tt.copy()
    bad[5] = bad[4]
    assert lib.bhip_proposal_create(h, dp(bad), 11, bh.MODEL_FHN, 2, dp(par), 5, C.byref(out)) == -1
    assert b"increasing" in lib.bhip_last_error(h)
    assert lib.bhip_proposal_create(h, dp(tt), 11, bh.MODEL_FHN, 2, dp(par), 5, C.byref(out)) == 0
    v = np.array([1.0])
    L = np.array([1.0, 0.0])
    assert lib.bhip_proposal_guide_lmmu(out, 1, dp(L), dp(v), None) == -4       # auxiliary not set yet
    assert lib.bhip_proposal_set_aux(out, bh.AUX_AFFINE, dp(np.zeros(3)), 3) == -1
    assert lib.bhip_proposal_set_aux(out, 7, dp(np.zeros(8)), 8) == -1
    assert lib.bhip_proposal_set_aux(out, bh.AUX_AFFINE, dp(np.array([0, 0, 1.0, 0, 0, 0, 0, 0.3])), 8) == 0
    assert lib.bhip_proposal_guide_lmmu(out, 3, dp(L), dp(v), None) == -1       # m > d
    assert lib.bhip_proposal_guide_lmmu(out, 1, dp(L), dp(v), None) == 0        # Sigma = 0 default: M[N] = Inf
    M = np.empty((11, 1))
    assert lib.bhip_proposal_guide_get(out, None, dp(M), None, None) == 0
    assert np.isinf(M[-1, 0]) and np.all(np.isfinite(M[:-1]))
    lib.bhip_proposal_destroy(out)
    with pytest.raises(bh.BridgeError, match="must be positive"):
        bh.OrnsteinUhlenbeck(-1.0, 1.0)


def test_welford_merge_equals_sequential_mcnext():
    # src/mclog.jl:31-56 semantics; merge = Chan's parallel form (used to combine per-GPU states)
    rng = np.random.default_rng(5)
    xs = rng.standard_normal((40, 6, 2))
    mc = bh.mcstart(xs[0])
    for x in xs:
        mc = bh.mcnext(mc, x)
    a = bh.mcstart(xs[0])
    b = bh.mcstart(xs[0])
    for x in xs[:15]:
        a = bh.mcnext(a, x)
    for x in xs[15:]:
        b = bh.mcnext(b, x)
    m, m2, n = bh.mcmerge(a, b)
    assert n == 40 and np.allclose(m, mc[0], atol=1e-14) and np.allclose(m2, mc[1], atol=1e-12)
    mean, cov = bh.mcstats(mc)
    assert np.allclose(cov[0], np.cov(xs[:, 0].T, ddof=1))
    lo, hi = bh.mcband(mc)
    assert np.all(lo < mean) and np.all(mean < hi)
    # oracle's mcnext agrees with the mirror's
    mo, m2o, no = np.zeros((6, 2)), np.zeros((6, 4)), 0
    for x in xs:
        no = o.mcnext(mo, m2o, no, x)
    assert np.allclose(mo, mc[0], atol=1e-15) and np.allclose(o.uncm(m2o, 2, 2), mc[1], atol=1e-13)


def test_coefficient_accessors_against_the_oracle():
    """bridgehip.b / sigma / a / Gamma / B / beta / r / H / guided_b (the reference's small accessor methods, evaluated on the
    host from the product's own parameters and guide arrays) against the oracle's restatement of the same methods"""
    h = bh.Context(-1)
    rng = np.random.default_rng(11)
    for c in problems.cases(41):
        P = c.bh_process(bh)
        Po = c.bh_proposal(bh, h)
        ref = c.oracle_proposal()
        for i in (0, 7, len(c.tt) - 2):
            x = np.asarray(c.x0) + 0.1 * rng.standard_normal(c.d)
            t = float(c.tt[i])
            assert np.allclose(bh.b(t, x, P), o.b(c.model, c.d, c.par, t, x), rtol=1e-14, atol=0), c.name
            assert np.allclose(bh.a(t, x, P), o.a(c.model, c.d, c.mp, c.par), rtol=1e-14, atol=0), c.name
            Pt = c.bh_aux(bh)
            assert np.allclose(bh.B(t, Pt), o.aux_B(c.aux, c.d, c.apar, t), rtol=1e-15, atol=0), c.name
            assert np.allclose(bh.beta(t, Pt), o.aux_beta(c.aux, c.d, c.apar, t), rtol=1e-14, atol=1e-300), c.name
            assert np.allclose(bh.a(t, Pt), o.aux_a(c.aux, c.d, c.mp, c.apar, t), rtol=1e-14, atol=0), c.name
            assert np.allclose(bh.b(t, x, Pt), o.aux_b(c.aux, c.d, c.apar, t, x), rtol=1e-12, atol=1e-14), c.name
            rr = o.guided_r(ref, i, x)
            assert np.allclose(bh.r(i, x, Po), rr, rtol=1e-9, atol=1e-12 * (1 + np.abs(rr).max())), c.name
            gb = o.guided_drift(ref, i, x)
            assert np.allclose(bh.guided_b(i, x, Po), gb, rtol=1e-9, atol=1e-12 * (1 + np.abs(gb).max())), c.name
        assert bh.constdiff(Po) and bh.constdiff(P)
        Hh = bh.H(3, Po)
        assert Hh.shape == (c.d, c.d) and np.allclose(Hh, Hh.T, rtol=1e-9, atol=1e-9 * np.abs(Hh).max())
    P = bh.OrnsteinUhlenbeck(2.0, 0.5)
    assert np.allclose(bh.Gamma(0.0, [0.3], P), [[4.0]]) and np.allclose(bh.sigma(0.0, [0.3], P), [[0.5]])
    with pytest.raises(bh.BridgeError):
        bh.b(0.0, [0.1, 0.2], bh.UserProcess(2, "o[0] = x[0]; o[1] = x[1];", [], [[1.0], [1.0]], ctx=h))


def test_bench_without_launcher_refuses_more_gpus_than_visible():
    """`python bench.py --gpus N` started bare drives the N devices from one process; beyond the visible devices (here, on
    a CPU box: none) it ends with rc 2, one clear line on stderr and no JSON -- never a "use a launcher" message"""
    import subprocess
    import sys
    import torch
    n = (torch.cuda.device_count() if torch.cuda.is_available() else 0) + 1
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 2, r.stderr[-1500:]
    assert f"--gpus {n} but only {n - 1} device(s) visible" in r.stderr and "torch.distributed.run" not in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_context_options_are_validated():
    """bhip_ctx_set_option on a host-only context: the value ranges of the round-4 options (BHIP_OPT_NOISE_SPEC: 4 | 3 | 2;
    BHIP_OPT_MID_VALU: 0, 1 or the largest dimension 4..12 that runs one path per lane), unknown options refused with a message"""
    c = bh.Context(-1)
    lib = c.lib
    for v in (3, 2, 4):
        assert lib.bhip_ctx_set_option(c.h, bh.OPT_NOISE_SPEC, v) == 0
    for v in (0, 1, 5, -2):
        assert lib.bhip_ctx_set_option(c.h, bh.OPT_NOISE_SPEC, v) == -1
        assert b"NOISE_SPEC" in lib.bhip_last_error(c.h)
    for v in (0, 1, 4, 8, 10, 12, 1):
        assert lib.bhip_ctx_set_option(c.h, bh.OPT_MID_VALU, v) == 0
    for v in (2, 3, 13, -1):
        assert lib.bhip_ctx_set_option(c.h, bh.OPT_MID_VALU, v) == -1
    for opt in (bh.OPT_WAVE_SPECIALISED, bh.OPT_TUNE_PLACEMENT, bh.OPT_FUSED_ARITHMETIC):
        assert lib.bhip_ctx_set_option(c.h, opt, 0) == 0 and lib.bhip_ctx_set_option(c.h, opt, 1) == 0
    assert lib.bhip_ctx_set_option(c.h, 99, 1) == -1 and b"unknown option" in lib.bhip_last_error(c.h)
    assert lib.bhip_ctx_set_option(None, bh.OPT_NOISE_SPEC, 2) == -1
    # bhip_ctx_get_option (round 6): what an option stands at -- the noise specification a stored run was drawn under
    fresh = bh.Context(-1)
    assert fresh.get_option(bh.OPT_NOISE_SPEC) == 4 and fresh.get_option(bh.OPT_FUSED_ARITHMETIC) == 0 and fresh.get_option(bh.OPT_WAVE_SPECIALISED) == 1
    fresh.set_option(bh.OPT_NOISE_SPEC, 2)
    fresh.set_option(bh.OPT_MID_VALU, 0)
    assert fresh.get_option(bh.OPT_NOISE_SPEC) == 2 and fresh.get_option(bh.OPT_MID_VALU) == 0
    fresh.set_option(bh.OPT_MID_VALU, 7)
    assert fresh.get_option(bh.OPT_MID_VALU) == 7
    import ctypes as C
    assert lib.bhip_ctx_get_option(fresh.h, 99, C.byref(C.c_int())) == -1 and lib.bhip_ctx_get_option(fresh.h, bh.OPT_NOISE_SPEC, None) == -1


def test_group_entry_points_reject_bad_arguments_without_a_device():
    lib = bh._lib.load()
    assert lib.bhip_chains_step_group(0, None, 0.9, 1, 0) == -1
    assert lib.bhip_chains_step_group(2, None, 0.9, 1, 0) == -1
    assert lib.bhip_chains_stats_group(1, None, None) == -1
    hs = (C.c_void_p * 2)(None, None)
    assert lib.bhip_chains_step_group(2, hs, 0.9, 1, 0) == -1
    assert lib.bhip_chains_step_group(65, hs, 0.9, 1, 0) == -1
