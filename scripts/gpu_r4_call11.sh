#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4k; mkdir -p $O
export TMPDIR=/tmp
cd $R
BRIDGEHIP_SO=$R/ab/exp.so timeout 1200 python scripts/gpu_arena_probe2.py 2>&1 | grep -v amdgpu.ids | tee $O/arena_probe2.txt
timeout 600 python bench.py --mode linpro32_mcmc --no-other-modes --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('linpro32_mcmc: kernel', round(r['kernel_avg_ms'],4), 'frac', round(r['frac'],4), d['config'].get('placement'))" | tee -a $O/tile.txt
