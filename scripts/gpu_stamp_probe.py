"""Round 4: the cycle budget of the wave-specialised kernels from s_memtime stamps (measurement builds of the library:
make EXTRA=-DPC_STAMP=1|2 OUT=../../ab/stampN.so BUILD=build_stampN, selected with BRIDGEHIP_SO).

  level 1 (PC_STAMP=1, production schedule): per wave, cycles spent working and cycles spent waiting at the hand-over barrier
  level 2 (PC_STAMP=2, consumer's interior loop one step per iteration, scheduling barriers at the marks): inside the consumer's
          step -- phase 0: loop overhead + W and the coefficient row in registers (LDS / scalar-load latency), 1: dw + issue of the X
          stores, 2: issue of the step's arithmetic, 3: wait for the new state, 4: an empty phase (= the cost of a mark)
"""
import ctypes as C
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
lib = ctx.lib
lib.bhip_debug_stamps.restype = C.c_int
lib.bhip_debug_stamps.argtypes = [C.c_void_p, C.c_size_t]
WORDS = 12 << 16
buf = np.zeros(WORDS, dtype=np.uint64)


def stamps():
    rc = lib.bhip_debug_stamps(buf.ctypes.data_as(C.c_void_p), WORDS)
    assert rc == 0, rc
    s = buf.reshape(-1, 12).astype(np.float64)
    return s[s[:, 0] > 0]


for mode in os.environ.get("STAMP_MODES", "c2 c2_fused proposals64k c4shard").split():
    chains = 0
    m = mode
    if mode == "proposals64k":
        m, chains = "proposals", 65536
    w = bench.Workload(m, ctx, chains, 0)
    ms = bench.kernel_times(w, 10, 3)
    stamps()                      # drop what the warm-up and the timing runs left
    w.step(); torch.cuda.synchronize()
    s = stamps()
    steps = bench.N_GRID - 1
    print(f"== {mode}: {w.kernel}  HIP events {np.mean(ms):.4f} ms per launch; {len(s)} waves stamped, cycles PER STEP (x{steps})")
    for role, name in ((1, "producer"), (2, "consumer")):
        r = s[s[:, 0] == role]
        if not len(r):
            continue
        tot, work, bar = r[:, 1] / steps, r[:, 2] / steps, r[:, 3] / steps
        print(f"  {name}: waves {len(r)}  total {tot.mean():7.1f} (min {tot.min():.1f} max {tot.max():.1f})  working {work.mean():7.1f}  "
              f"at the barrier {bar.mean():7.1f}")
        if role == 2 and r[:, 4:].sum() > 0:
            ph = r[:, 4:10].mean(axis=0) / steps
            print("     level 2, per step: " + "  ".join(f"ph{k} {ph[k]:.1f}" for k in range(5)))
    del w
    torch.cuda.empty_cache()
