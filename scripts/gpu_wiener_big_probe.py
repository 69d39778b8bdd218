import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import numpy as np, torch, bridgehip as bh
ctx = bh.Context(0)
tt = np.linspace(0, 1, 1001)
for mp, n in ((32, 65536), (30, 65536), (8, 262144), (5, 262144), (12, 131072), (16, 131072)):
    W = bh.EnsemblePath(tt, mp, n, ctx, parts=1)
    bh.sample_(W, bh.Wiener(mp), seed=1); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): bh.sample_(W, bh.Wiener(mp), seed=1)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print(f"sample!(W, Wiener{{{mp}}}) {n} paths x 1000: {ms:.3f} ms = {8.0 * mp * n * 1000 / ms / 1e6:.0f} GB/s")
    del W
