"""Does the headline kernel's time depend on WHERE its buffers were allocated?  Same process, the workload is created, timed and
destroyed several times, optionally with a dummy allocation of varying size in front of it (shifts the physical placement)."""
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
pads = [0, 0, 0, 1 << 30, 3 << 30, (5 << 30) + (1 << 21), 0, 7 << 30, 0, 0][:int(os.environ.get('PROBE_REPS', '10'))]
for rep, pad in enumerate(pads):
    dummy = torch.empty(pad, dtype=torch.uint8, device=ctx.device) if pad else None
    w = bench.Workload(mode, ctx, 0, 0)
    ms = bench.kernel_times(w, 30, 5)
    free, total = torch.cuda.mem_get_info()
    ptr = None
    if w.chains is not None:
        import ctypes as C
        p, ld = C.c_void_p(), C.c_long()
        ctx.lib.bhip_chains_proposal_X(w.chains.h, C.byref(p), C.byref(ld))
        ptr = p.value
    print(f"rep {rep}: pad {pad >> 20:6d} MiB  mean {np.mean(ms):.4f} ms  min {np.min(ms):.4f}  Xo at {ptr if ptr is None else hex(ptr)}  free {free >> 20} MiB", flush=True)
    del w, dummy
    import gc; gc.collect()
    torch.cuda.empty_cache()
    torch.cuda.synchronize()
