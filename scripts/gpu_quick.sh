# quick GPU check: parity tests + both bench modes (one line each)
mkdir -p gpurun_out
(timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -25) > gpurun_out/pytest_gpu.log; tail -3 gpurun_out/pytest_gpu.log
for m in mcmc proposals; do timeout 300 python bench.py --steps 20 --warmup 3 --mode $m --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default', d['config']['mode'], '%.3e'%d['value'], 'ms', round(d['roofline']['kernel_avg_ms'],3), 'frac', round(d['roofline']['frac'],3), d['config'].get('acceptance_rate'))"; done
