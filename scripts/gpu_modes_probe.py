"""Kernel times of the secondary modes on the bench workload (262 144 paths x 1000 steps): external-W solve, stand-alone
llikelihood, innovations, girsanov, stand-alone Wiener sampling, chain re-materialisation."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bench
import bridgehip as bh

ctx = bh.default_context(0)
P, N = 262144, bench.N_GRID
Po = bench.build_proposal(ctx)
x0 = np.array(bench.X0)


def timeit(fn, n=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


W = bh.sample(Po.tt, bh.Wiener(1), npaths=P, seed=1, ctx=ctx)
X = bh.EnsemblePath(Po.tt, 2, P, ctx)
ll = ctx.empty(P)
rows = []
rows.append(("sample!(W, Wiener()) alone", timeit(lambda: bh.sample_(W, bh.Wiener(1), seed=1)), 8))
rows.append(("solve!(X, x0, W, Po) + ll, external W", timeit(lambda: bh.solve_(bh.Euler(), X, x0, W, Po, ll=ll)), 24))
rows.append(("llikelihood(LeftRule(), X, Po) alone", timeit(lambda: bh.llikelihood(bh.LeftRule(), X, Po)), 16))
P2 = bh.FitzHughNagumo(0.1, 0.0, 1.5, 0.8, 0.3, 0.4)
proc2 = bh.PlainProcess(Po.tt, P2, ctx=ctx)
X2, _, _ = bh.sample_solve(x0, proc2, P, seed=2)
W2 = bh.EnsemblePath(Po.tt, 2, P, ctx)
rows.append(("innovations!(W, X, P) (Models.FitzHughNagumo)", timeit(lambda: bh.innovations_(bh.EulerMaruyama(), W2, X2, proc2)), 32))
rows.append(("girsanov(X, P, Pt)", timeit(lambda: bh.girsanov(X2, proc2, bh.FitzHughNagumo(0.12, 0.0, 1.4, 0.8, 0.3, 0.4))), 16))
ch = bh.Chains(Po, x0, P, seed=3)
ch.step(0.9, 2)
rows.append(("chains.current_X() (gather W + solve)", timeit(lambda: ch.current_X(), 3), 16 + 8 + 8 + 16))
for name, ms, b in rows:
    print(f"{name:52s} {ms:8.3f} ms   {b * P * (N - 1) / ms / 1e6:7.0f} GB/s of {b} B/path-step")
