#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4u; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_pc.py tests/test_gpu_parity.py tests/test_gpu_noise.py tests/test_gpu_fused.py tests/test_gpu_segchains.py tests/test_gpu_adapt_device.py -m gpu -x -q 2>&1 | tail -3
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $O/small_modes.txt
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import bench, bridgehip as bh
ctx = bh.default_context(0)
for mode, P in (("c2", 0), ("c4shard", 0), ("proposals", 65536), ("proposals", 32768), ("nclar", 65536), ("c2_fused", 0), ("mcmc", 65536)):
    w = bench.Workload(mode, ctx, P, 0)
    for turn in range(2):
        ms = bench.kernel_times(w, 30, 5, min_ms=100.0)
        print(f"{mode:>10} P {w.P:>6}: mean {np.mean(ms):.4f} ms  min {np.min(ms):.4f}  frac {w.P*1000*w.bytes_per_pathstep/np.mean(ms)/1e6/8000:.3f}", flush=True)
    del w
PY
