#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
MODES="mcmc nclar_mcmc c4shard" bash scripts/gpu_profile_all.sh > gpurun_out/r4_profile_chains.log 2>&1
tail -12 gpurun_out/r4_profile_chains.log | cut -c1-200
BENCH_SAME_DEVICE=1 timeout 900 python bench.py --gpus 8 --steps 10 --warmup 2 > gpurun_out/r4_bench_8ctx_same_device.json 2> gpurun_out/r4_bench_8ctx.err
python -c "
import json; d=json.load(open('gpurun_out/r4_bench_8ctx_same_device.json')); print({k: d.get(k) for k in ('n_gpus','ms_per_step','host_issue_us_per_step','per_gpu_ms_per_step')}); print(d.get('survey_c4',{}).get('host_issue_us_per_step'), d.get('survey_c4',{}).get('ms_per_step'))"
