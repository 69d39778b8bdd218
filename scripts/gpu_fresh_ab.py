"""Round 4: large ensembles of FRESH proposals on the wave-specialised kernel (k_pc<.., 6, .., 1>: the normals in
a producer wave, 8 workgroups of 128 threads per CU) against the one-lane kernel k_paths, same box, alternating.
BHIP_PC_FRESH_MAX moves the switch-over (default 98 304 paths)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np

import bench
import bridgehip as bh

ctx = bh.default_context(0)
for mode in os.environ.get("PROBE_MODES", "proposals nclar c2").split():
    for P in (262144, 131072):
        w = bench.Workload(mode, ctx, P, 0)
        for turn in range(2):
            for name, mx in (("k_paths", "0"), ("k_pc", "100000000")):
                os.environ["BHIP_PC_FRESH_MAX"] = mx
                ms = bench.kernel_times(w, 20, 3, min_ms=60.0)
                frac = P * 1000 * w.bytes_per_pathstep / (np.mean(ms) * 1e-3) / 1e9 / 8000
                print(f"{mode:>10} P {P:>7} {name:>8}: mean {np.mean(ms):.4f} ms  min {np.min(ms):.4f}  frac {frac:.3f}", flush=True)
        os.environ.pop("BHIP_PC_FRESH_MAX", None)
        del w
