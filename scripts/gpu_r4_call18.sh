#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4r; mkdir -p $O
cd $R
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $O/shard_place.txt
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, time
import bench, bridgehip as bh
ctx = bh.default_context(0)
for tune in (1, 0, 1, 0):
    ctx.set_option(bh.OPT_TUNE_PLACEMENT, tune)
    ws = []
    for k in range(4):
        w = bench.Workload("c4shard", ctx, 0, 0)
        ws.append(w)
        ms = bench.kernel_times(w, 40, 5, min_ms=100.0)
        p = w.chains.placement()
        print(f"c4shard placement {tune} ensemble {k}: {np.mean(ms):.4f} ms per launch (min {np.min(ms):.4f}), pairs {p['tries']}, reference {p['ms_first']:.4f}, kept {p['ms_best']:.4f}", flush=True)
    del ws
for P in (65536, 131072):
    for tune in (1, 0):
        ctx.set_option(bh.OPT_TUNE_PLACEMENT, tune)
        w = bench.Workload("mcmc", ctx, P, 0)
        ms = bench.kernel_times(w, 30, 5, min_ms=100.0)
        p = w.chains.placement()
        print(f"mcmc {P} chains placement {tune}: {np.mean(ms):.4f} ms, pairs {p['tries']}, reference {p['ms_first']:.4f}, kept {p['ms_best']:.4f}  frac {P*1000*32/np.mean(ms)/1e6/8000:.3f}", flush=True)
        del w
PY
