#!/bin/bash
# Round 4: the rank-aware placement in the product -- GPU suite, fresh-process bench lines, several ensembles in one process
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4j; mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; grep -E "passed|failed|rror" $O/pytest.txt | tail -3
for i in 1 2 3; do
  timeout 600 python bench.py --no-other-modes --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('fresh process $i: ms_per_step', round(d['ms_per_step'],4), 'kernel', round(r['kernel_avg_ms'],4), 'frac', round(r['frac'],4), d['config']['placement'])" | tee -a $O/bench_fresh.txt
done
for m in nclar_mcmc linpro32_mcmc c4shard; do
  timeout 600 python bench.py --mode $m --no-other-modes --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$m: kernel', round(r['kernel_avg_ms'],4), 'frac', round(r['frac'],4), d['config'].get('placement'))" | tee -a $O/bench_fresh.txt
done
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $O/multi.txt
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, time
import bench, bridgehip as bh
ctx = bh.default_context(0)
ws = []
for k in range(6):
    t0 = time.perf_counter()
    w = bench.Workload("mcmc", ctx, 0, 0)
    dt = time.perf_counter() - t0
    ws.append(w)
    ms = bench.kernel_times(w, 20, 3)
    print(f"ensemble {k}: set-up {dt*1e3:.0f} ms, {np.mean(ms):.4f} ms per launch, placement {w.chains.placement()}", flush=True)
PY
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json
