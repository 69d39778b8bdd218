#!/bin/bash
# same-box A/B of bench modes: the current build vs an older library build (default ab/r2.so = the round-2 library built from
# its commit), alternating, two rounds.  usage: [AB_OLD=ab/x.so] gpu_ab3.sh mode...
set -o pipefail   # a step's exit code is its command's, not that of the `tail` / `tee` behind it
OLD=${AB_OLD:-ab/r2.so}
for rep in 1 2; do
  for v in old cur; do
    so=""; ob=0; [ $v = old ] && so=$PWD/$OLD && ob=1
    for m in "$@"; do
      BRIDGEHIP_SO_OLD_BUILD=$ob BRIDGEHIP_SO=$so python bench.py --mode $m --steps 10 --warmup 3 --no-cpu-baseline --no-other-modes 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['config']['mode'], round(d['roofline']['kernel_avg_ms'],4), round(d['roofline']['kernel_min_ms'],4), round(d['roofline']['frac'],4))"
    done
  done
done
