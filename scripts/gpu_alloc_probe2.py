"""Allocation vs time: several workloads ALIVE at once, timed in turns with idle gaps.  If a workload's time is a property of its
allocation it stays with it; if it is a property of the moment, all of them move together."""
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
mode = os.environ.get("PROBE_MODE", "mcmc")
ws = {}


def t(name):
    ms = bench.kernel_times(ws[name], 30, 3)
    print(f"  {name}: mean {np.mean(ms):.4f}  min {np.min(ms):.4f}", flush=True)


for rnd in range(3):
    for name in "ABCDE":
        if name not in ws:
            ws[name] = bench.Workload(mode, ctx, 0, 0)
            print(f"allocated {name}", flush=True)
        t(name)
        time.sleep(0.3)
    print("-- again, reverse order")
    for name in "EDCBA":
        t(name)
    if rnd == 0:
        for name in "BD":
            del ws[name]
        gc.collect(); torch.cuda.empty_cache()
        print("freed B, D (re-allocated next round)")
