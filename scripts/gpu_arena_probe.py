"""Round 4: W and Xo at explicit offsets inside ONE process-wide physically contiguous block (BHIP_PLACE=arena:<GiB>:<W MiB>:<Xo MiB>):
the physical frame is fixed for the whole sweep, so the pCN kernel's time as a function of where its two regions lie can be read
off -- absolute positions or their distance?"""
import gc
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch

import bench
import bridgehip as bh

ctx = bh.default_context(0)
ctx.set_option(bh.OPT_TUNE_PLACEMENT, 0)
mode = os.environ.get("PROBE_MODE", "mcmc")
SZ = int(os.environ.get("ARENA_GIB", "200"))
G = 1024


def run(wo, xo, tag):
    os.environ["BHIP_PLACE"] = f"arena:{SZ}:{int(wo * G)}:{int(xo * G)}"
    try:
        w = bench.Workload(mode, ctx, 0, 0)
    except Exception as e:   # noqa: BLE001
        print(f"{tag} W at {wo:>6} GiB, Xo at {xo:>6} GiB: FAILED {str(e)[:100]}", flush=True)
        return
    ms = bench.kernel_times(w, 12, 3)
    print(f"{tag} W at {wo:>6} GiB, Xo at {xo:>6} GiB (distance {xo - wo:>7}): mean {np.mean(ms):.4f} ms  min {np.min(ms):.4f}", flush=True)
    del w
    gc.collect(); torch.cuda.empty_cache(); torch.cuda.synchronize()


top = SZ - 5
for xo in [5] + list(range(8, top, 8)):
    run(0, xo, "A")                      # W fixed at the bottom
for wo in list(range(0, top, 8)):
    if abs(wo - 100) >= 5:
        run(wo, 100, "B")                # Xo fixed in the middle
for a in range(0, top - 5, 12):
    run(a, a + 5, "C")                   # neighbours, moving together
for a in range(0, top - 32, 16):
    run(a, a + 32, "D")                  # 32 GiB apart, moving together
