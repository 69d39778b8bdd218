"""Round 4: the physical memory of an MI355X is three regions of 96 GiB (profiles/r4_placement_regions.txt); the pCN kernel is fast when
its chain lines and its proposal paths lie in DIFFERENT regions.  Recipe: W, a transient spacer, Xo (BHIP_PLACE=spacer:<GiB>): does the
allocator place them so?  Several ensembles alive at once, several spacer sizes, plain and contiguous allocations."""
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
for mode in os.environ.get("PROBE_MODES", "mcmc").split():
    for spec in os.environ.get("PROBE_SPECS", "spacer:96 spacer:96:contig spacer:64 spacer:100 spacer:128 spacer:48 spacer:24 malloc").split():
        if spec == "malloc":
            os.environ.pop("BHIP_PLACE", None)
        else:
            os.environ["BHIP_PLACE"] = spec
        ws, line = [], []
        for r in range(5):
            w = bench.Workload(mode, ctx, 0, 0)
            ws.append(w)
            ms = bench.kernel_times(w, 16, 3)
            free, _ = torch.cuda.mem_get_info()
            line.append(f"{np.mean(ms):.4f}")
        print(f"{mode} {spec:>18}: " + "  ".join(line) + f"   (free after the fifth: {free >> 30} GiB)", flush=True)
        del ws
        gc.collect(); torch.cuda.empty_cache(); torch.cuda.synchronize()
