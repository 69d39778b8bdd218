"""Handles may be destroyed in any order (Python's collector at interpreter exit and Julia finalizers give no order):
the context is reference counted by its children (bhip_ctx_destroy only closes it while they live)."""
import subprocess
import sys
import os

import numpy as np
import pytest

import bridgehip as bh

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_context_destroyed_before_its_children():
    ctx = bh.Context(0)
    tt = np.linspace(0.0, 1.0, 65)
    P = bh.LinPro([[-0.5]], [0.0], [[0.8]])
    Po = bh.GuidedBridge(tt, P, P, [0.3], ctx=ctx)
    ch = bh.Chains(Po, [0.1], 256, seed=1)
    ch.step(0.9, 2)
    h_ctx, h_po, h_ch = ctx.h, Po.h, ch.h
    L = ctx.lib
    # the context first, then the chains, then the proposal: each call must be safe
    L.bhip_ctx_destroy(h_ctx); ctx.h = None
    assert L.bhip_chains_step(h_ch, 0.9, 1, 0) == -4                    # BHIP_ESTATE: no new work through a destroyed context ...
    L.bhip_chains_destroy(h_ch); ch.h = None                             # ... but its children remain destroyable
    L.bhip_proposal_destroy(h_po); Po.h = None


@pytest.mark.parametrize("mode", ["linpro32_mcmc", "c4shard"])
def test_interpreter_exit_with_live_handles_is_clean(mode):
    """bench.py leaves its workload alive until the interpreter finalises: exit code 0, nothing on stderr but the driver note"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--mode", mode, "--steps", "1", "--warmup", "1", "--chains", "4096",
                        "--no-cpu-baseline", "--no-other-modes"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "terminate called" not in r.stderr and '"metric"' in r.stdout


def test_readme_quickstart_runs():
    env = dict(os.environ, QUICKSTART_PATHS="2048")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "quickstart.py")], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "acceptance" in r.stdout, r.stdout + r.stderr[-2000:]


def test_bhip_free_of_a_foreign_or_freed_pointer_does_not_release_the_context():
    """advisor r2: bhip_free released the context's reference for ANY pointer; a double free could free the context under
    its live children"""
    import ctypes as C
    ctx = bh.Context(0)
    L = ctx.lib
    p = C.c_void_p()
    assert L.bhip_malloc(ctx.h, 1024, C.byref(p)) == 0
    assert L.bhip_free(ctx.h, p) == 0
    assert L.bhip_free(ctx.h, p) == -1 and b"double free" in L.bhip_last_error(ctx.h)      # BHIP_EINVAL, reference count untouched
    assert L.bhip_free(ctx.h, C.c_void_p(0x1000)) == -1
    tt = np.linspace(0.0, 1.0, 33)
    P = bh.LinPro([[-0.5]], [0.0], [[0.8]])
    Po = bh.GuidedBridge(tt, P, P, [0.3], ctx=ctx)
    ch = bh.Chains(Po, [0.1], 128, seed=1)
    ch.step(0.9, 1)                                           # the context is alive and usable
    assert np.isfinite(ch.ll()).all()
