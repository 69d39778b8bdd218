mkdir -p gpurun_out
(timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -25) > gpurun_out/pytest_gpu.log; tail -4 gpurun_out/pytest_gpu.log
for m in mcmc proposals; do timeout 300 python bench.py --steps 10 --warmup 2 --mode $m --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default', d['config']['mode'], '%.3e'%d['value'], 'ms', round(d['roofline']['kernel_avg_ms'],3), 'frac', round(d['roofline']['frac'],3))"; done
for v in pf2 pf8 w6 w8; do BRIDGEHIP_SO=$PWD/bridge.jl_amd/variants/lib_$v.so timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['config']['mode'], '%.3e'%d['value'], 'ms', round(d['roofline']['kernel_avg_ms'],3), 'frac', round(d['roofline']['frac'],3))"; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r1 -o mcmc -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_r1.log 2>&1
ls $GRAFT_REPO_ROOT/gpurun_out/prof_r1 | head
