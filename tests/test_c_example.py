"""examples/fhn_chains.c drives libbridgehip.so from plain C (no Python, no PyTorch in that process): the C ABI is the
product boundary.  CPU part: it compiles and links against the header and the library.  GPU part: its output equals
what the Python mirror computes for the same seeds, bit for bit."""
import os
import re
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path, name="fhn_chains"):
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("gcc not available")
    exe = str(tmp_path / name)
    lib = os.path.join(ROOT, "bridge.jl_amd")
    subprocess.check_call([gcc, "-O2", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", name + ".c"),
                           "-L", lib, "-lbridgehip", "-Wl,-rpath," + lib, "-lm", "-o", exe])
    return exe


@pytest.mark.parametrize("name", ["fhn_chains", "lorenz_smoothing", "fhn_chains_multi"])
def test_c_example_compiles_against_the_header(tmp_path, name):
    exe = _build(tmp_path, name)
    assert os.path.exists(exe)


@pytest.mark.gpu
def test_c_example_matches_the_python_mirror(tmp_path):
    import bridgehip as bh
    import problems
    exe = _build(tmp_path)
    nchains, iters = 300, 6
    out = subprocess.run([exe, str(nchains), str(iters)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    got = {int(m.group(1)): (int(m.group(2)), float.fromhex(m.group(3)))
           for m in re.finditer(r"chain (\d+) acc (\d+) ll (\S+)", out.stdout)}
    assert len(got) == 4
    ctx = bh.default_context(0)
    P = bh.FitzhughDiffusion(0.1, 0.0, 1.5, 0.8, 0.3)
    Po = bh.PartialBridge(problems.tau_grid(2.0, 1001), P, bh.fitzhugh_aux_linearised_end(P, 1.1), [[1.0, 0.0]], [1.1], [[1e-10]], ctx=ctx)
    ch = bh.Chains(Po, [-0.5, -0.6], nchains, seed=44, store_X=False)
    ch.step(0.9, iters)
    ll, acc = ch.ll(), ch.acc()
    for p, (a, l) in got.items():
        assert a == acc[p] and l == ll[p], (p, a, acc[p], l, ll[p])
    m = re.search(r"acceptance ([0-9.]+) mean ll (\S+)", out.stdout)
    assert abs(float(m.group(1)) - acc.sum() / (nchains * iters)) < 1e-4 and abs(float(m.group(2)) - ll.mean()) < 1e-5
    assert abs(float(re.search(r"endpoint x1 (\S+)", out.stdout).group(1)) - 1.1) < 1e-3


@pytest.mark.gpu
def test_c_smoothing_example_matches_the_python_mirror(tmp_path):
    """examples/lorenz_smoothing.c: the adaptive smoothing loop (chained LinearAppr segments, joint MH, mcnext!, per-chain
    adaptation on the device) driven through the C ABI from plain C == the same loop through the Python mirror, bit for bit"""
    import math
    import bridgehip as bh
    exe = _build(tmp_path, "lorenz_smoothing")
    n, iters, adaptit = 200, 12, 5
    out = subprocess.run([exe, str(n), str(iters), str(adaptit)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    got = {int(m.group(1)): (int(m.group(2)), float.fromhex(m.group(3))) for m in re.finditer(r"chain (\d+) acc (\d+) ll (\S+)", out.stdout)}
    assert len(got) == 4
    ctx = bh.default_context(0)
    m, M = 3, 40
    par = ((10.0, 20.0, 8.0 / 3), (3.0, 3.0, 3.0))
    tgrid = np.array([0.24 * i / (m * M) for i in range(m * M + 1)])
    ref = np.zeros((m * M + 1, 3)); y = np.array([1.5, -1.5, 25.0])
    for i in range(m * M + 1):
        ref[i] = y
        if i < m * M:
            b = np.array([10.0 * (y[1] - y[0]), y[0] * (20.0 - y[2]) - y[1], y[0] * y[1] - 8.0 / 3 * y[2]])
            y = y + b * (tgrid[i + 1] - tgrid[i])
    obs = np.array([[ref[j * M, k] + 0.3 * ((j + k) % 3 - 1) for k in range(3)] for j in range(m + 1)])
    L, Sig = np.eye(3), 0.25 * np.eye(3)
    P = bh.Lorenz(*par)
    HT, vT = bh.gpupdate(1e3 * np.eye(3), np.zeros(3), L, Sig, obs[m])
    H, v, segs = HT, vT, [None] * m
    for i in range(m - 1, -1, -1):
        segs[i] = bh.GuidedBridge(tgrid[i * M:(i + 1) * M + 1].copy(), P, bh.linearappr(ref[i * M:(i + 1) * M + 1]), v, H, ctx=ctx)
        H, v = bh.gpupdate(segs[i], L, Sig, obs[i])
    C = np.zeros((3, 3))
    C[0, 0] = math.sqrt(H[0, 0]); C[1, 0] = H[0, 1] / C[0, 0]; C[1, 1] = math.sqrt(H[1, 1] - C[1, 0] * C[1, 0])
    C[2, 0] = H[0, 2] / C[0, 0]; C[2, 1] = (H[1, 2] - C[1, 0] * C[2, 0]) / C[1, 1]; C[2, 2] = math.sqrt(H[2, 2] - C[2, 0] * C[2, 0] - C[2, 1] * C[2, 1])
    sc = bh.SegChains(segs, v, C, n, seed=7, mcnext=True)
    w_new, w_old = math.sqrt(0.1), math.sqrt(0.9)
    for it in range(1, iters + 1):
        if it % adaptit == 0:
            sc.adapt_device(L, Sig, obs[:m], HT, vT, newblock=True, doaccept=(it == adaptit))
        sc.step(w_old, w_new, 1)
    ll, acc, _ = sc.state()
    for p, (a, l) in got.items():
        s = 0.0
        for i in range(m):
            s += ll[i, p]
        assert a == acc[p] and l == s, (p, a, acc[p], l, s)


@pytest.mark.gpu
def test_c_multi_gpu_example_equals_one_ensemble_with_all_the_chains(tmp_path):
    """examples/fhn_chains_multi.c: ONE plain-C process drives every visible device through bhip_comm_init_all +
    bhip_comm_allgather_group (on a one-GPU test box: a world of one through the same calls).  Device k owns the global
    chain ids [k*n, (k+1)*n), so its chains and the combined statistics equal ONE ensemble of ndev*n chains, bit for bit."""
    import bridgehip as bh
    import problems
    exe = _build(tmp_path, "fhn_chains_multi")
    n, iters = 320, 5
    out = subprocess.run([exe, str(n), str(iters)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    ndev = int(re.search(r"devices (\d+)", out.stdout).group(1))
    assert ndev == bh._lib.load().bhip_device_count()
    got = {int(m.group(1)): (int(m.group(2)), float.fromhex(m.group(3))) for m in re.finditer(r"chain (\d+) acc (\d+) ll (\S+)", out.stdout)}
    assert sorted(got) == sorted(k * n + p for k in range(ndev) for p in range(2))
    ctx = bh.default_context(0)
    P = bh.FitzhughDiffusion(0.1, 0.0, 1.5, 0.8, 0.3)
    Po = bh.PartialBridge(problems.tau_grid(2.0, 1001), P, bh.fitzhugh_aux_linearised_end(P, 1.1), [[1.0, 0.0]], [1.1], [[1e-10]], ctx=ctx)
    ch = bh.Chains(Po, [-0.5, -0.6], ndev * n, seed=44, store_X=False)
    ch.step(0.9, iters)
    ll, acc = ch.ll(), ch.acc()
    for p, (a, l) in got.items():
        assert a == acc[p] and l == ll[p], (p, a, acc[p], l, ll[p])
    m = re.search(r"chains (\d+) iterations (\d+) acceptance ([0-9.]+) mean ll (\S+)", out.stdout)
    assert int(m.group(1)) == ndev * n and int(m.group(2)) == iters
    assert abs(float(m.group(3)) - acc.sum() / (ndev * n * iters)) < 1e-6 and abs(float(m.group(4)) - ll.mean()) < 1e-7 * max(1.0, abs(ll.mean()))
    # asking for more devices than the box has is refused up front
    r = subprocess.run([exe, "64", "1", str(ndev + 1)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 2 and "visible" in r.stderr
