#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4l; mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_gpu_tile.py tests/test_gpu_usertile.py tests/test_gpu_segchains.py tests/test_gpu_noise.py -m gpu -x -q 2>&1 | tail -3
for rep in 1 2; do
 for v in pair old; do
  so=""; [ $v = old ] && so=$R/ab/tl0.so
  for m in linpro32_mcmc; do
   BRIDGEHIP_SO=$so timeout 600 python bench.py --mode $m --steps 10 --warmup 3 --no-other-modes --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$v $m: kernel', round(r['kernel_avg_ms'],4), 'min', round(r['kernel_min_ms'],4), 'frac', round(r['frac'],4))" | tee -a $O/tl_ab.txt
  done
 done
done
python -m pytest tests/test_gpu_group.py -m gpu -q -s -k host_issue 2>&1 | grep -E "us of host|passed|failed" | tee $O/host_issue.txt
