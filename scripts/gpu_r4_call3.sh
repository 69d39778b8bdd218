#!/bin/bash
# Round 4, third GPU call: W-Xo distance sweep inside contiguous blocks; fresh proposals k_pc vs k_paths at large sizes; live traffic pass
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4c; mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 1200 python scripts/gpu_delta_probe.py 2>&1 | grep -v amdgpu.ids > $O/delta_probe.txt; tail -30 $O/delta_probe.txt
PROBE_MODE=nclar_mcmc PROBE_GAPS="0 64 256 1024 2048 4096 8192 0" timeout 600 python scripts/gpu_delta_probe.py 2>&1 | grep -v amdgpu.ids > $O/delta_probe_nclar.txt; cat $O/delta_probe_nclar.txt
timeout 900 python scripts/gpu_fresh_ab.py 2>&1 | grep -v amdgpu.ids > $O/fresh_ab.txt; cat $O/fresh_ab.txt
timeout 600 python bench.py --no-other-modes --no-cpu-baseline > $O/bench_quick.json 2> $O/bench_quick.err; python -c "
import json; d=json.load(open('$O/bench_quick.json')); r=d['roofline']; print({k: r.get(k) for k in ('kernel_avg_ms','frac','traffic','traffic_box','traffic_over_algorithmic','traffic_live_failed','traffic_source')})"
