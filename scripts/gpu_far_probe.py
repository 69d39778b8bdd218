"""Round 4: the pCN kernel is fast (1.52 ms) when the chain lines W and the proposal paths Xo lie >= 32 GiB apart inside ONE physically
contiguous block, slow (1.78 ms) at every distance up to 16 GiB (profiles/r4_placement_*.txt).  Map it: explicit offsets of W and Xo
in a contiguous block (BHIP_PLACE=contig2:<W MiB>:<Xo MiB>) -- is it the distance or the absolute position?"""
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
G = 1024
pairs = [(0, 5), (0, 17), (0, 20), (0, 22), (0, 24), (0, 26), (0, 28), (0, 30), (0, 31), (0, 32), (0, 34), (0, 36), (0, 40), (0, 48), (0, 64), (0, 96), (0, 128), (0, 192),
         (32, 37), (64, 69), (36, 0), (32, 0), (24, 0), (16, 48), (8, 40), (20, 52), (48, 80), (100, 132), (100, 116)]
if os.environ.get("PROBE_PAIRS"):
    pairs = [tuple(float(x) for x in q.split(":")) for q in os.environ["PROBE_PAIRS"].split()]
for (wo, xo) in pairs:
    os.environ["BHIP_PLACE"] = f"contig2:{int(wo * G)}:{int(xo * G)}"
    try:
        w = bench.Workload(mode, ctx, 0, 0)
    except Exception as e:   # noqa: BLE001
        print(f"W at {wo:>6} GiB, Xo at {xo:>6} GiB: FAILED {str(e)[:80]}", flush=True)
        continue
    ms = bench.kernel_times(w, 16, 3)
    print(f"{mode} W at {wo:>6} GiB, Xo at {xo:>6} GiB (distance {xo - wo:>7}): mean {np.mean(ms):.4f} ms  min {np.min(ms):.4f}", flush=True)
    del w
    gc.collect(); torch.cuda.empty_cache(); torch.cuda.synchronize()
