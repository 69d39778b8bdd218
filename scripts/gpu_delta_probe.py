"""Round 4: in the headline kernel the read cursor in the chain lines and the write cursor in the proposal paths advance at the SAME
rate (16 ld bytes per step for d = 2, m' = 1): the distance D between them is constant for the whole launch.  If the allocation
lottery (1.48 vs 1.76 ms) is a function of D modulo some period of the memory system, a physically CONTIGUOUS allocation
(hipDeviceMallocContiguous) lets the library choose D.  Sweep the W-Xo gap inside one contiguous block (BHIP_PLACE=contig:<KiB>)."""
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
gaps = [int(g) for g in os.environ.get("PROBE_GAPS", "").split()] or (
    [0] + [64 * k for k in range(1, 33)] + [2048 + 256 * k for k in range(1, 25)] + [8192 * k for k in range(1, 17)] + [0, 64, 1024])
keep = []
for gap in gaps:
    os.environ["BHIP_PLACE"] = f"contig:{gap}"
    try:
        w = bench.Workload(mode, ctx, 0, 0)
    except Exception as e:   # noqa: BLE001
        print(f"gap {gap:>7} KiB: FAILED {e}", flush=True)
        break
    ms = bench.kernel_times(w, 16, 3)
    print(f"gap {gap:>7} KiB: mean {np.mean(ms):.4f} ms  min {np.min(ms):.4f}", flush=True)
    del w
    gc.collect(); torch.cuda.empty_cache(); torch.cuda.synchronize()
# the same gap in blocks that land elsewhere: is the figure a property of the gap or still of the block?
for gap in (0, 1024):
    os.environ["BHIP_PLACE"] = f"contig:{gap}"
    ws = []
    for r in range(4):
        w = bench.Workload(mode, ctx, 0, 0)
        ws.append(w)
        ms = bench.kernel_times(w, 16, 3)
        print(f"gap {gap:>7} KiB, block {r} (all alive): mean {np.mean(ms):.4f} ms", flush=True)
    del ws
    gc.collect(); torch.cuda.empty_cache(); torch.cuda.synchronize()
