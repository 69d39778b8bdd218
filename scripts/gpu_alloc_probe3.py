"""Which of the pCN kernel's streams makes its time depend on the allocation?  Five ensembles alive at once, each timed with and
without the proposal-path store (bhip_chains_step with iters = 2: only the last iteration stores Xo; iters = 1: every launch does),
and the fresh-proposal kernel (write-only X) on five separately allocated ensembles."""
import gc
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch

import bench
import bridgehip as bh

ctx = bh.default_context(0)
P = 262144
Po = bench.build_proposal(ctx)


def timeit(fn, n=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


chs = [bh.Chains(Po, np.array(bench.X0), P, seed=4, store_X=True) for _ in range(5)]
for k, ch in enumerate(chs):
    t1 = timeit(lambda: ch.step(0.9, 1))
    t8 = timeit(lambda: ch.step(0.9, 8), 3) / 8
    print(f"chains {k}: every launch stores Xo {t1:.4f} ms   8 iterations per call (one Xo store) {t8:.4f} ms per iteration", flush=True)
del chs
gc.collect()
chs = [bh.Chains(Po, np.array(bench.X0), P, seed=4, store_X=False) for _ in range(5)]
for k, ch in enumerate(chs):
    print(f"chains without Xo {k}: {timeit(lambda: ch.step(0.9, 1)):.4f} ms", flush=True)
del chs
gc.collect()
ws = [bench.Workload("proposals", ctx, 0, 0) for _ in range(5)]
for k, w in enumerate(ws):
    print(f"proposals {k}: {timeit(w.step):.4f} ms", flush=True)
