#!/usr/bin/env python3
"""The reference-shaped call sequence on the device: sample!(W) -> solve!(Euler, X, W, Po) with llikelihood fused -> stand-alone
llikelihood(LeftRule, X, Po), each a launch of its own, FHN PartialBridge 262 144 paths x 1001 grid points.  ms per call (HIP events)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bridgehip as bh
import bench

ctx = bh.Context(0)
Po = bench.build_proposal(ctx)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
W = bh.sample(Po.tt, bh.Wiener(1), npaths=n, seed=1, ctx=ctx)
X = bh.EnsemblePath(Po.tt, 2, n, ctx)
ll = ctx.empty(n)

def t(fn, k=10):
    for _ in range(3):
        fn()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(k + 1)]
    for i in range(k):
        e[i].record(); fn()
    e[k].record(); torch.cuda.synchronize()
    ts = [e[i].elapsed_time(e[i + 1]) for i in range(k)]
    return sum(ts) / k, min(ts)

steps = len(Po.tt) - 1
for name, fn, byt in (("sample!(W)", lambda: bh.sample_(W, bh.Wiener(1), seed=1), 8),
                      ("solve!(X, W, Po) + ll", lambda: bh.solve_(bh.EulerMaruyama(), X, bench.X0, W, Po, ll=ll), 24),
                      ("solve!(X, W, Po)", lambda: bh.solve_(bh.EulerMaruyama(), X, bench.X0, W, Po), 24),
                      ("llikelihood(X, Po)", lambda: bh.llikelihood(bh.LeftRule(), X, Po), 16)):
    a, m = t(fn)
    print(f"{name:26s} {a:8.4f} ms (min {m:.4f})   {byt} B/path-step -> {byt * n * steps / a / 1e6:8.1f} GB/s = {byt * n * steps / a / 1e6 / 8000:.3f} of 8 TB/s")
