#!/bin/bash
# Round-2 profile set: every bench mode through scripts/gpu_profile.sh (kernel trace + separate FETCH / WRITE / SQ passes).
# Summaries land in gpurun_out/r2_<mode>_{trace,fetch,write,sq,sq2}.txt; copy them into profiles/.
for m in ${MODES:-mcmc c4shard c2 proposals nclar nclar_mcmc linpro32 linpro32_mcmc}; do
  bash scripts/gpu_profile.sh r2_$m --mode $m --steps 6 --warmup 2 --no-cpu-baseline --no-other-modes > /dev/null 2>&1
  grep -h "k_pc\|k_paths\|k_tile\|k_chain" gpurun_out/r2_${m}_trace.txt | head -2 | cut -c1-170
done
