#!/bin/bash
# Round 4, closing call: the GPU suite, smoke(), the default bench line twice (fresh processes), the launcher form at one rank
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4z; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; grep -E "passed|failed|rror" $O/pytest.txt | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for i in 1 2; do
  ( time timeout 900 python bench.py > $O/bench_$i.json 2> $O/bench_$i.err ) 2>&1 | grep real
  python -c "
import json; d=json.load(open('$O/bench_$i.json')); r=d['roofline']; print('bench $i: value %.4e ms_per_step %.4f kernel %.4f frac %.4f' % (d['value'], d['ms_per_step'], r['kernel_avg_ms'], r['frac']), d['config']['placement']['tries'], r.get('traffic_box'), round(r.get('traffic_over_algorithmic') or 0, 4), 'host_issue_us', round(d.get('host_issue_us_per_step', 0), 1)); print(json.dumps(d['modes']))"
done
WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 BENCH_PER_RANK=1 timeout 600 python bench.py --steps 10 --warmup 2 --no-other-modes --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('launcher form, one rank:', round(d['ms_per_step'],4), d['config']['launch'][:90])"
