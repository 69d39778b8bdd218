#!/usr/bin/env python3
"""Lorenz-96 (component-wise user drift, hipRTC) GuidedBridge at d = 4, 5, 8: one path per lane (k_paths<MUser>) vs the zero-padded MFMA
tile kernel (BHIP_OPT_MID_VALU = 0).  262 144 fresh proposals x 1000 steps, ms per launch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bridgehip as bh

L96 = "o = (x[(k+1)%d] - x[(k+d-2)%d])*x[(k+d-1)%d] - x[k] + par[0];"
ctx = bh.Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144

def t(fn, k=5):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k

for d in (4, 5, 8):
    rng = np.random.default_rng(d)
    sig = 0.4 * np.eye(d) + 0.05 * rng.standard_normal((d, d)) / np.sqrt(d)
    tt = np.linspace(0.0, 0.5, 1001)
    x0 = 2.0 + 0.3 * rng.standard_normal(d); v = 2.0 + 0.3 * rng.standard_normal(d)
    P = bh.UserProcessComponents(d, L96, [2.0], sig, ctx=ctx)
    Po = bh.GuidedBridge(tt, P, bh.LinPro(-np.eye(d), 2.0 * np.ones(d), sig), v, ctx=ctx)
    X = bh.EnsemblePath(tt, d, n, ctx); ll = ctx.empty(n)
    out = []
    for opt in (1, 0):
        ctx.set_option(bh.OPT_MID_VALU, opt)
        out.append(t(lambda: bh.sample_solve(x0, Po, n, seed=1, store_X=True)))
    ctx.set_option(bh.OPT_MID_VALU, 1)
    print(f"d = {d}: one path per lane {out[0]:8.3f} ms   zero padded on the tile kernel {out[1]:8.3f} ms   x{out[1] / out[0]:.2f}")
