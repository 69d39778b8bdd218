#!/bin/bash
# Round-3 profile set: every bench mode through scripts/gpu_profile.sh (kernel trace + separate FETCH / WRITE / SQ passes), then the
# default bench line ON THE SAME BOX (VERDICT r2 hygiene: traffic / valu and kernel_avg_ms from one machine).
# Summaries land in gpurun_out/r3_<mode>_{trace,fetch,write,sq,sq2}.txt; copy them into profiles/.
# (chain modes: 30 steps, so that the handful of launches of the placement tuning weigh little in the averages)
for m in ${MODES:-mcmc c4shard c2 proposals nclar nclar_mcmc linpro4 linpro32 linpro32_mcmc}; do
  steps=8; case $m in mcmc|c4shard|nclar_mcmc) steps=30;; esac
  bash scripts/gpu_profile.sh r3_$m --mode $m --steps $steps --warmup 2 --no-cpu-baseline --no-other-modes > /dev/null 2>&1
  grep -h "k_pc\|k_paths\|k_tile\|k_chain" gpurun_out/r3_${m}_trace.txt | head -2 | cut -c1-200
done
if [ -z "$NO_BENCH" ]; then
  mkdir -p profiles && cp gpurun_out/r3_*_{trace,fetch,write,sq,sq2}.txt profiles/ 2>/dev/null
  python bench.py > gpurun_out/r3_bench_samebox.json 2> gpurun_out/r3_bench_samebox.err
fi
# the reference-shaped three-call sequence (sample! / solve! / llikelihood as separate launches), same box
python scripts/gpu_ext_probe.py 2>/dev/null | grep "B/path-step" > gpurun_out/r3_separate_calls.txt
