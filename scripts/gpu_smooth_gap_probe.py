#!/usr/bin/env python3
"""Is the smoothing iteration's time the sum of its kernels?  One bhip_segchains_step call of 20 iterations against 20 calls of one."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import bench, bridgehip as bh
side = torch.cuda.Stream() if os.environ.get('PROBE_SIDE_STREAM') else None
ctx = bh.Context(0, stream=side.cuda_stream) if side else bh.Context(0)
m, M, n = 4, 250, 32768
P = bh.Lorenz((10.0, 20.0, 8 / 3), (3.0, 3.0, 3.0))
tgrid = np.linspace(0.0, 0.002 * m * M, m * M + 1)
Y = np.zeros((m * M + 1, 3)); y = np.array([1.5, -1.5, 25.0])
for i in range(m * M + 1):
    Y[i] = y
    if i < m * M:
        y = y + np.array([10 * (y[1] - y[0]), y[0] * (20 - y[2]) - y[1], y[0] * y[1] - 8 / 3 * y[2]]) * (tgrid[i + 1] - tgrid[i])
L, Sig = np.eye(3), 0.25 * np.eye(3)
obs = Y[::M] + 0.5 * np.random.default_rng(0).standard_normal((m + 1, 3))
H, v = bh.gpupdate(1e3 * np.eye(3), np.zeros(3), L, Sig, obs[m])
segs = [None] * m
for i in range(m - 1, -1, -1):
    segs[i] = bh.GuidedBridge(tgrid[i * M:(i + 1) * M + 1].copy(), P, bh.linearappr(Y[i * M:(i + 1) * M + 1]), v, H, ctx=ctx)
    H, v = bh.gpupdate(segs[i], L, Sig, obs[i])
sc = bh.SegChains(segs, v, bh.cholupper_t(H), n, seed=1, mcnext=True)
wo, wn = 0.9, math.sqrt(1 - 0.81)
K = 20
sc.step(wo, wn, 3); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(side); sc.step(np.full(K, wo), np.full(K, wn)); e1.record(side); torch.cuda.synchronize()
print("one call of %d iterations: %.3f ms per iteration" % (K, e0.elapsed_time(e1) / K))
e0.record(side)
for _ in range(K):
    sc.step(wo, wn, 1)
e1.record(side); torch.cuda.synchronize()
print("%d calls of one iteration: %.3f ms per iteration" % (K, e0.elapsed_time(e1) / K))
