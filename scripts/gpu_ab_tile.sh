#!/bin/bash
# same-box A/B of the d = 32 kernels: the current build against bridge.jl_amd/variants/head.so (the last commit), alternating
set -o pipefail   # a step's exit code is its command's, not that of the `tail` / `tee` behind it
for rep in 1 2; do
  for v in head cur; do
    so=""; [ $v = head ] && so=$PWD/bridge.jl_amd/variants/head.so
    for m in linpro32 linpro32_mcmc; do
      BRIDGEHIP_SO=$so python bench.py --mode $m --steps 6 --warmup 2 --no-cpu-baseline --no-other-modes 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['config']['mode'], round(d['roofline']['kernel_avg_ms'],3), round(d['roofline']['frac'],4))"
    done
  done
done
