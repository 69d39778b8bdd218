# A/B: every library under bridge.jl_amd/variants + the default build, both bench modes
set -o pipefail   # a step's exit code is its command's, not that of the `tail` / `tee` behind it
for so in default $(ls bridge.jl_amd/variants/*.so); do
  for m in mcmc proposals; do
    if [ $so = default ]; then unset BRIDGEHIP_SO; else export BRIDGEHIP_SO=$PWD/$so; fi
    timeout 300 python bench.py --steps 10 --warmup 2 --mode $m --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$so', d['config']['mode'], '%.3e'%d['value'], 'avg_ms', round(r['kernel_avg_ms'],3), 'frac', round(r['frac'],3))"
  done
done
