"""Round 4: physically contiguous allocations (hipDeviceMallocContiguous) are ALWAYS the slow kind (1.78 ms per pCN iteration against
1.48 on the lucky hipMalloc ones, profiles/r4_placement_*.txt): the kernel's strides are powers of two (2 MiB per path row, 64 MiB per
chunk of chain lines at 262 144 chains), and on contiguous physical memory that camps on a subset of the DRAM banks.  Sweep the
leading dimension (BHIP_LD_PAD chains, multiples of 64) under contiguous and under plain allocations."""
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
pads = [int(g) for g in os.environ.get("PROBE_PADS", "0 64 128 192 256 320 448 576 704 1088 2112 4160 8256 16448 32832 1024 4096 16384 65536").split()]
for place in os.environ.get("PROBE_PLACES", "contig malloc").split():
    if place == "malloc":
        os.environ.pop("BHIP_PLACE", None)
    else:
        os.environ["BHIP_PLACE"] = place
    for pad in pads:
        os.environ["BHIP_LD_PAD"] = str(pad)
        ws, line = [], []
        for r in range(3):
            w = bench.Workload(mode, ctx, 0, 0)
            ws.append(w)
            ms = bench.kernel_times(w, 16, 3)
            line.append(f"{np.mean(ms):.4f}")
        print(f"{mode} {place:>8} ld pad {pad:>6}: " + "  ".join(line), flush=True)
        del ws
        gc.collect(); torch.cuda.empty_cache(); torch.cuda.synchronize()
os.environ.pop("BHIP_LD_PAD", None)
