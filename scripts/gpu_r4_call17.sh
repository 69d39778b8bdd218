#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4q; mkdir -p $O
cd $R
timeout 600 ./ab/hbm_rank_probe 2>&1 | tee $O/hbm_rank_probe.txt
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $O/multi.txt
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, time
import bench, bridgehip as bh
ctx = bh.default_context(0)
ws = []
for k in range(8):
    t0 = time.perf_counter()
    w = bench.Workload("mcmc" if k != 5 else "mcmc_v2noise", ctx, 0, 0)
    dt = time.perf_counter() - t0
    ws.append(w)
    ms = bench.kernel_times(w, 20, 3)
    p = w.chains.placement()
    print(f"ensemble {k}: set-up {dt*1e3:.0f} ms, {np.mean(ms):.4f} ms per launch, pairs timed {p['tries']}, same-piece reference {p['ms_first']:.3f}, kept {p['ms_best']:.3f}", flush=True)
PY
timeout 900 python -m pytest tests/test_gpu_pc.py tests/test_gpu_parity.py tests/test_gpu_lifetime.py -m gpu -x -q 2>&1 | tail -2
