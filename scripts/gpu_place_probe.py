"""Round 4: does the WAY the ensemble's memory is obtained decide which of the two populations (1.55-1.61 vs 1.75-1.83 ms per pCN
iteration, profiles/r3_alloc_placement.txt) an allocation falls into?  hipMalloc against the virtual-memory API with a chosen
alignment of the virtual range, chosen physical chunking and a chosen W-Xo offset (BHIP_PLACE, bhip_api.hip chains_alloc_state).
Placement tuning off; REPS ensembles per specification, all alive at once (so that each lands elsewhere), timed in turn."""
import gc
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import ctypes as C

import numpy as np
import torch

import bench
import bridgehip as bh

ctx = bh.default_context(0)
ctx.set_option(bh.OPT_TUNE_PLACEMENT, 0)
mode = os.environ.get("PROBE_MODE", "mcmc")
reps = int(os.environ.get("PROBE_REPS", "4"))
specs = os.environ.get("PROBE_SPECS", "malloc vmm:0:0 vmm:2:0 vmm:1024:0 vmm:4096:0 vmm:32768:0 vmm:2:2 vmm:1024:1024 vmm:1024:2 "
                                      "vmm:1024:0:64 vmm:1024:0:256 vmm:1024:0:1024 malloc").split()


def xo_ptr(w):
    p, ld = C.c_void_p(), C.c_long()
    ctx.lib.bhip_chains_proposal_X(w.chains.h, C.byref(p), C.byref(ld))
    return p.value


for spec in specs:
    if spec == "malloc":
        os.environ.pop("BHIP_PLACE", None)
    else:
        os.environ["BHIP_PLACE"] = spec
    ws = []
    for r in range(reps):
        try:
            w = bench.Workload(mode, ctx, 0, 0)
        except Exception as e:   # noqa: BLE001
            print(f"{spec:>18} rep {r}: FAILED {e}", flush=True)
            break
        ws.append(w)
        ms = bench.kernel_times(w, 24, 4)
        print(f"{spec:>18} rep {r}: mean {np.mean(ms):.4f} ms  min {np.min(ms):.4f}  Xo at {xo_ptr(w):#x}", flush=True)
    # a second turn in reverse order: the figure stays with the allocation
    for r in range(len(ws) - 1, -1, -1):
        ms = bench.kernel_times(ws[r], 24, 2)
        print(f"{spec:>18} rep {r} again: mean {np.mean(ms):.4f} ms  min {np.min(ms):.4f}", flush=True)
    del ws
    gc.collect(); torch.cuda.empty_cache(); torch.cuda.synchronize()
