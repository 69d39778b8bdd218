"""Probe: d=32 tile kernel with in-kernel noise vs external W (no Philox/Box-Muller VALU work).
Tells whether fp64 MFMA and fp64/int VALU overlap on gfx950 or serialise."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bridgehip as bh

d, N, P = 32, 1001, 65536
ctx = bh.default_context(0)
rng = np.random.default_rng(5)
G, G2 = rng.standard_normal((d, d)) / np.sqrt(d), rng.standard_normal((d, d)) / np.sqrt(d)
sig = 0.5 * np.eye(d) + 0.05 * G2
Po = bh.GuidedBridge(np.linspace(0.0, 1.0, N), bh.LinPro(-np.eye(d) + 0.1 * G, np.zeros(d), sig),
                     bh.LinPro(-np.eye(d), np.zeros(d), sig), 0.5 * np.ones(d), ctx=ctx)
x0 = np.zeros(d)
W = bh.sample(Po.tt, bh.Wiener(d), npaths=P, seed=1, ctx=ctx)
X = bh.EnsemblePath(Po.tt, d, P, ctx)
ll = ctx.empty(P)


def timeit(fn, n=3):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


t_ext = timeit(lambda: bh.solve_(bh.Euler(), X, x0, W, Po, ll=ll))
t_ext_nox = timeit(lambda: ctx.check(ctx.lib.bhip_solve(ctx.h, Po.h, bh.api._dptr(x0), None, W.ptr(), P, None, P, bh.api.vp(ll.data_ptr()), 0, P)))
t_fresh = timeit(lambda: ctx.check(ctx.lib.bhip_sample_solve(ctx.h, Po.h, bh.api._dptr(x0), None, None, P, X.ptr(), P,
                                                              bh.api.vp(ll.data_ptr()), 0, P, 5, 1, 0)))
t_fresh_nox = timeit(lambda: ctx.check(ctx.lib.bhip_sample_solve(ctx.h, Po.h, bh.api._dptr(x0), None, None, P, None, P,
                                                                  bh.api.vp(ll.data_ptr()), 0, P, 5, 1, 0)))
t_w = timeit(lambda: bh.sample_(W, bh.Wiener(d), seed=1))
flops = P * (N - 1) * 10240
for name, t in (("ext W + X store", t_ext), ("ext W, ll only", t_ext_nox), ("fresh noise + X store", t_fresh),
                ("fresh noise, ll only", t_fresh_nox), ("k_wiener_big alone (32 normals/step)", t_w)):
    print(f"{name:40s} {t:8.3f} ms   {flops / t / 1e9:7.2f} TFLOP/s-equivalent")
