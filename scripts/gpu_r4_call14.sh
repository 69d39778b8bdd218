#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4n; mkdir -p $O
export TMPDIR=/tmp
cd $R
PROBE_DIMS="4 5 6 8" timeout 900 python scripts/gpu_midchain_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/midchain_probe.txt
PROBE_DIMS="4 6 8 9 10 11 12" timeout 900 python scripts/gpu_mid_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/mid_probe.txt
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; grep -E "passed|failed|rror" $O/pytest.txt | tail -3
for m in mcmc proposals c2 nclar linpro4 linpro32; do
  timeout 600 python bench.py --mode $m --no-other-modes --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$m: kernel', round(r['kernel_avg_ms'],4), 'frac', round(r['frac'],4))" | tee -a $O/modes.txt
done
