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


def _build(tmp_path):
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("gcc not available")
    exe = str(tmp_path / "fhn_chains")
    lib = os.path.join(ROOT, "bridge.jl_amd")
    subprocess.check_call([gcc, "-O2", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "fhn_chains.c"),
                           "-L", lib, "-lbridgehip", "-Wl,-rpath," + lib, "-lm", "-o", exe])
    return exe


def test_c_example_compiles_against_the_header(tmp_path):
    exe = _build(tmp_path)
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
