#!/bin/bash
# The headline's 262 144 chains as 1, 2, 3 and 4 shards on ONE device (one context + stream each, BENCH_SAME_DEVICE test double of the N > 1
# path; chains keyed by global id, so the values are those of the one ensemble): does dealing the chain state over more pieces of the
# device memory -- every shard places its own W and Xo apart -- and running the shards' launches concurrently move the headline?
#     gpurun -- 'bash scripts/gpu_shards_same_device.sh <tag>'
set -o pipefail   # a step's exit code is its command's, not that of the `tail` / `tee` behind it
TAG=${1:-shards}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
COMMON="--mode mcmc --no-cpu-baseline --no-live-traffic --no-other-modes --steps 100 --warmup 5"
python bench.py $COMMON > $OUT/n1.json 2> $OUT/n1.err
for spec in "2 131072" "3 87424" "4 65536"; do
  set -- $spec
  BENCH_SAME_DEVICE=1 python bench.py --gpus $1 --chains $2 $COMMON > $OUT/n$1.json 2> $OUT/n$1.err
done
TAGV=$TAG python - <<'P' > $OUT/summary.txt
import json, glob, os
for f in sorted(glob.glob(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", ""+os.environ.get('TAGV','shards')+"", "n*.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), "contexts", d["n_gpus"], "chains each", d["config"]["paths_per_gpu"], "ms/step", round(d["ms_per_step"], 4),
              "value", f'{d["value"]:.4g}', "per-context ms", [round(x, 4) for x in d.get("per_gpu_ms_per_step", [])])
    except Exception as e:
        print(f, "unreadable:", e)
P
cat $OUT/summary.txt
