"""Round 4: issue priorities of the small-ensemble wave-specialised kernels (BHIP_TUNE: bit 0 no consumer priority, bit 1 producer
priority).  The s_memtime stamps say the PRODUCER wave is the critical one at C2 (383 of 397 cycles per step working, the consumer
waits 180 at the hand-over barrier): does the priority of round 3 (consumer first) still pay?"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np

import bench
import bridgehip as bh

ctx = bh.default_context(0)
for mode, P in (("c2", 0), ("c4shard", 0), ("proposals", 65536), ("proposals", 32768), ("nclar", 65536)):
    w = bench.Workload(mode, ctx, P, 0)
    for turn in range(2):
        for t in ("0", "1", "2", "3"):
            os.environ["BHIP_TUNE"] = t
            ms = bench.kernel_times(w, 30, 5, min_ms=60.0)
            print(f"{mode:>10} P {w.P:>6} tune {t}: mean {np.mean(ms):.4f} ms  min {np.min(ms):.4f}", flush=True)
    os.environ.pop("BHIP_TUNE", None)
    del w
