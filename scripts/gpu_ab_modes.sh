#!/bin/bash
# same-box A/B of arbitrary bench modes: current build vs bridge.jl_amd/variants/$OLD.so, alternating.  usage: gpu_ab_modes.sh <old> mode...
set -o pipefail   # a step's exit code is its command's, not that of the `tail` / `tee` behind it
OLD=$1; shift
for rep in 1 2; do
  for v in old cur; do
    so=""; ob=0; [ $v = old ] && so=$PWD/bridge.jl_amd/variants/$OLD.so && ob=1
    for m in "$@"; do
      BRIDGEHIP_SO_OLD_BUILD=$ob BRIDGEHIP_SO=$so python bench.py --mode $m --steps 8 --warmup 3 --no-cpu-baseline --no-other-modes 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['config']['mode'], round(d['roofline']['kernel_avg_ms'],4), round(d['roofline']['frac'],4))"
    done
  done
done
