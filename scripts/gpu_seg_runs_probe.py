"""Round 4: the smoothing loop's commit + mcnext! pass writes three streams (Xc, mean, m2).  With the second moments of all segments in ONE
contiguous run and means + current paths in another (BHIP_SEG_RUNS=1), does the iteration time become a property of where the two runs
landed -- as the pCN kernel's did?  Several ensembles alive at once, timed in turn; plain allocations for comparison."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import math

import numpy as np
import torch

import bench
import bridgehip as bh

ctx = bh.default_context(0)
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
HT, vT = bh.gpupdate(1e3 * np.eye(3), np.zeros(3), L, Sig, obs[m])
H, v, segs = HT, vT, [None] * m
for i in range(m - 1, -1, -1):
    segs[i] = bh.GuidedBridge(tgrid[i * M:(i + 1) * M + 1].copy(), P, bh.linearappr(Y[i * M:(i + 1) * M + 1]), v, H, ctx=ctx)
    H, v = bh.gpupdate(segs[i], L, Sig, obs[i])
wo, wn = 0.9, math.sqrt(1 - 0.81)


def t(sc, k=10):
    sc.step(wo, wn, 2); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); sc.step(wo, wn, k); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k


for runs in ("1", "0", "1"):
    os.environ["BHIP_SEG_RUNS"] = runs
    scs, line = [], []
    for r in range(6):
        sc = bh.SegChains(segs, v, bh.cholupper_t(H), n, seed=1, mcnext=True)
        scs.append(sc)
        line.append(f"{t(sc):.4f}")
    line2 = [f"{t(sc):.4f}" for sc in scs]
    print(f"BHIP_SEG_RUNS={runs}: " + "  ".join(line) + "   again: " + "  ".join(line2), flush=True)
    del scs
    torch.cuda.empty_cache()
