"""One fresh process: the default container of the `proposals` mode (262 144 paths x 1001 x 2: two buffers of 2.1 GB) -- how many of its
buffers bhip_alloc_apart found pairwise apart, how long the allocation took, and the kernel time of the fused proposal into it.
Run several times in a row (fresh processes see different allocator states):  for i in 1 2 3 4 5 6; do python scripts/gpu_apart_probe.py; done"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch

import bench
import bridgehip as bh

ctx = bh.Context(0)
torch.cuda.synchronize()
t0 = time.perf_counter()
w = bench.Workload(sys.argv[1] if len(sys.argv) > 1 else "proposals", ctx, 0, 0)
torch.cuda.synchronize()
t1 = time.perf_counter()
ms = bench.kernel_times(w, 20, 3, min_ms=100.0)
print(f"{w.mode}: buffers {w.nparts}, pairwise apart {w.parts_apart}, set-up {1e3 * (t1 - t0):.0f} ms, kernel {ms.avg:.4f} ms = {w.roofline(ms)['frac']:.3f}", flush=True)
