"""pCN chains without the per-iteration proposal-path store (BHIP_CHAINS_STORE_X off): the chain state (W, ll, parity) is
complete without it and the current X is re-materialised on demand; bench workload, 262 144 chains."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bench
import bridgehip as bh

ctx = bh.default_context(0)
P = 262144
Po = bench.build_proposal(ctx)
for store in (True, False):
    ch = bh.Chains(Po, bench.X0, P, seed=4, store_X=store)
    ch.step(0.9, 3)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ch.step(0.9, 20)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"store_X={store}: {ms:.3f} ms per iteration, {P * 1000 / ms / 1e6:.1f}e9 path-steps/s, acceptance {ch.acc().sum() / (P * 23):.3f}")
    del ch
