"""Round 4, run under `rocprofv3 --kernel-trace --pmc ...` (one counter set per run): NENS ensembles of the bench workload alive
at once (hipMalloc, placement tuning off), LAUNCHES pCN iterations on each in turn.  The dispatches of k_pc come in blocks of
LAUNCHES + 1 per ensemble (one warm-up); scripts/rocpd_dispatches.py lists duration and counters per dispatch, so a slow and a
fast allocation can be compared counter by counter within ONE process.  Prints the HIP-event figure of every block."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np

import bench
import bridgehip as bh

ctx = bh.default_context(0)
ctx.set_option(bh.OPT_TUNE_PLACEMENT, 0)
nens = int(os.environ.get("PMC_NENS", "5"))
launches = int(os.environ.get("PMC_LAUNCHES", "6"))
places = os.environ.get("PMC_PLACES", "").split()     # per ensemble: malloc | contig | vmm:... (BHIP_PLACE); default: hipMalloc for all
ws = []
for k in range(nens):
    pl = places[k] if k < len(places) else "malloc"
    if pl == "malloc":
        os.environ.pop("BHIP_PLACE", None)
    else:
        os.environ["BHIP_PLACE"] = pl
    ws.append(bench.Workload(os.environ.get("PROBE_MODE", "mcmc"), ctx, 0, 0))
for k, w in enumerate(ws):
    ms = bench.kernel_times(w, launches, 1)
    print(f"ensemble {k}: mean {np.mean(ms):.4f} ms  min {np.min(ms):.4f}", flush=True)
