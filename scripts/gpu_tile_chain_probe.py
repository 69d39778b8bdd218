"""d = 32 pCN chains on the tile kernel (slot layout): time per iteration, 65 536 chains x 1000 steps."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bench
import bridgehip as bh

ctx = bh.default_context(0)
w = bench.Workload("linpro32", ctx, 262144, 0)
for store in (True, False):
    ch = bh.Chains(w.Po, np.zeros(32), w.P, seed=5, store_X=store)
    ch.step(0.95, 2)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ch.step(0.95, 1)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"d=32 chains, store_X={store}: {ms:.2f} ms per iteration, {w.P * 1000 / ms / 1e6:.2f}e9 path-steps/s, "
          f"{w.P * 1000 * 10240 / ms / 1e9:.1f} TFLOP/s-equivalent, acceptance {ch.acc().sum() / (w.P * 7):.3f}")
    del ch
