"""Round 4: does the workgroup -> chain-group mapping (bhip_path_kernel.h xcd_block; the hardware deals workgroup b to XCD b % 8,
every XCD has its own L2 with address-interleaved channels) decide the two populations of allocations?  For every mapping
(BHIP_XCD_MAP = 0 identity, 1 rotated, 2 contiguous per XCD) REPS ensembles of every mode, all alive, timed in turn; then the same
ensembles again under the other mappings (the mapping is read per launch: same allocation, different mapping)."""
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
reps = int(os.environ.get("PROBE_REPS", "4"))
for mode in os.environ.get("PROBE_MODES", "mcmc proposals nclar_mcmc c4shard c2").split():
    os.environ["BHIP_XCD_MAP"] = "0"
    ws = [bench.Workload(mode, ctx, 0, 0) for _ in range(reps)]
    for turn in range(2):
        for m in ("0", "1", "2"):
            os.environ["BHIP_XCD_MAP"] = m
            line = []
            for w in ws:
                ms = bench.kernel_times(w, 24, 3, min_ms=30.0)
                line.append(f"{np.mean(ms):.4f}")
            print(f"{mode:>12} map {m} turn {turn}: " + "  ".join(line), flush=True)
    del ws
    gc.collect(); torch.cuda.empty_cache(); torch.cuda.synchronize()
