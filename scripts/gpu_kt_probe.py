"""ms per launch of bench modes by bench.kernel_times (>= 100 ms of back-to-back launches after a burst of the same length: the
protocol of the default run's `other_modes`):   python scripts/gpu_kt_probe.py c2 c2_fused proposals:65536 c4shard ...
Prints mode, paths, ms (back to back), min / max of the per-launch pass, fraction of the mode's roofline."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch

import bench
import bridgehip as bh

ctx = bh.Context(0)
for item in sys.argv[1:]:
    mode, _, P = item.partition(":")
    w = bench.Workload(mode, ctx, int(P) if P else 0, 0)
    ms = bench.kernel_times(w, 20, 3, min_ms=100.0)
    r = w.roofline(ms)
    print(f"{item:28s} {w.P:8d} paths  {ms.avg:9.4f} ms  [{min(ms):.4f} .. {max(ms):.4f}]  frac {r['frac']:.3f} ({r['bound']})  {w.kernel}", flush=True)
    if getattr(w, "nparts", 1) > 1:
        w.X.free()
    del w
    torch.cuda.empty_cache()
