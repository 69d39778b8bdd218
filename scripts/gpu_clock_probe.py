"""What a freshly acquired box does to the headline kernel over its first minutes: ms per launch of the bench default (HIP events,
bursts of 40 launches) next to the clocks rocm-smi reports, from a cold start.  (Round 3: the same box ran the same kernel at 1.83 ms
and, a minute later, at 1.61 ms -- gpurun_out/r3c/ab.txt.)"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch

import bench
import bridgehip as bh


def clocks():
    try:
        r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp"], capture_output=True, text=True, timeout=20)
        out = []
        for line in r.stdout.splitlines():
            for key in ("sclk", "mclk", "fclk", "socclk", "Power (W)", "Socket Power", "(junction)", "(memory)"):
                if key in line:
                    out.append(line.split(":", 1)[-1].strip().replace("clock level", "").replace("  ", " "))
        return " | ".join(out)
    except Exception as e:
        return f"rocm-smi: {e}"


t00 = time.perf_counter()
print("before any GPU work:", clocks(), flush=True)
ctx = bh.default_context(0)
w = bench.Workload(os.environ.get("PROBE_MODE", "mcmc"), ctx, 0, 0)
print(f"setup done at {time.perf_counter() - t00:.1f} s", flush=True)
idle = float(os.environ.get("PROBE_IDLE", "0"))
for burst in range(int(os.environ.get("PROBE_BURSTS", "40"))):
    ms = bench.kernel_times(w, 40, 0)
    line = f"t={time.perf_counter() - t00:6.1f}s  burst {burst:2d}: mean {np.mean(ms):.4f} ms  min {np.min(ms):.4f}  max {np.max(ms):.4f}"
    if burst % 4 == 0:
        line += "   " + clocks()
    print(line, flush=True)
    if idle:
        time.sleep(idle)
