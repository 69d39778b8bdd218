"""Round 4: inside ONE process-wide contiguous 200-GiB block (BHIP_PLACE=arena:..., library built with -DBHIP_PLACE_EXPERIMENTS): locate the
first cut between two 96-GiB pieces by bisection with the pCN kernel itself, then time W and Xo in arrangements that put a region
ASTRIDE a cut -- does spreading one stream over two pieces (twice the banks) beat clean separation?  (The luckiest plain hipMalloc
mixtures reach 1.45-1.48 ms, the clean separation 1.53.)"""
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
SZ, G = 200, 1024


def run(wo, xo, tag=None, n=12):
    os.environ["BHIP_PLACE"] = f"arena:{SZ}:{int(wo * G)}:{int(xo * G)}"
    w = bench.Workload(mode, ctx, 0, 0)
    ms = float(np.mean(bench.kernel_times(w, n, 3)))
    del w
    gc.collect(); torch.cuda.empty_cache(); torch.cuda.synchronize()
    if tag:
        print(f"{tag}: W at {wo:8.3f} GiB, Xo at {xo:8.3f} GiB: {ms:.4f} ms", flush=True)
    return ms


t_same, t_far = run(0, 5), run(0, 100)
print(f"same piece {t_same:.4f} ms, far apart {t_far:.4f} ms", flush=True)
thr = 0.5 * (t_same + t_far)
lo, hi = 5.0, 100.0            # Xo at lo: slow (same piece as W at 0); at hi: fast
while hi - lo > 0.26:
    mid = round((lo + hi) / 2 * 4) / 4
    if run(0, mid, n=8) > thr:
        lo = mid
    else:
        hi = mid
# Xo = [x, x + 4] GiB: slow while most of it lies below the cut -> the cut is near lo + 2
c1 = lo + 2.0
c2 = c1 + 96.0
print(f"first cut near {c1:.2f} GiB (Xo at {lo} slow, at {hi} fast), second expected near {c2:.2f}", flush=True)
h = 2.0
run(c1 - 7, c1 + 2, "clean: W piece 0, Xo piece 1      ")
run(c1 - 7, c2 + 2, "clean: W piece 0, Xo piece 2      ")
run(c1 - h, c2 + 2, "W astride cut 1, Xo piece 2       ")
run(c1 - 7, c2 - h, "W piece 0, Xo astride cut 2       ")
run(c1 - h, c2 - h, "W astride cut 1, Xo astride cut 2 ")
run(c1 - h, c1 + 8, "W astride cut 1, Xo piece 1       ")
run(c1 - 7, c1 - h, "W piece 0, Xo astride cut 1       ")
run(c1 - 1.0, c2 + 2, "W 1/4 : 3/4 over cut 1, Xo piece 2")
run(c1 - 7, c1 + 2, "clean again                       ")
