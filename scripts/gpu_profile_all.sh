#!/bin/bash
# Profile set of a round (T=r4 by default; round 3: T=r3): every bench mode through scripts/gpu_profile.sh (kernel trace + separate FETCH / WRITE / SQ passes), then the
# default bench line ON THE SAME BOX (VERDICT r2 hygiene: traffic / valu and kernel_avg_ms from one machine).
# Summaries land in gpurun_out/${T}_<mode>_{trace,fetch,write,sq,sq2}.txt; copy them into profiles/.
# (large chain modes: 100 steps, so that the <= 17 launches of the placement -- the same-piece reference among them, slow by construction --
# weigh little in the averages; the summaries carry the median too)
T=${T:-r4}
for m in ${MODES:-mcmc c4shard c2 proposals nclar nclar_mcmc linpro4 linpro32 linpro32_mcmc}; do
  steps=8; case $m in mcmc|nclar_mcmc) steps=100;; c4shard) steps=30;; esac
  bash scripts/gpu_profile.sh ${T}_$m --mode $m --steps $steps --warmup 2 --no-cpu-baseline --no-other-modes > /dev/null 2>&1
  grep -h "k_pc\|k_paths\|k_tile\|k_chain" gpurun_out/${T}_${m}_trace.txt | head -2 | cut -c1-200
done
if [ -z "$NO_BENCH" ]; then
  mkdir -p profiles && cp gpurun_out/${T}_*_{trace,fetch,write,sq,sq2}.txt profiles/ 2>/dev/null
  python bench.py > gpurun_out/${T}_bench_samebox.json 2> gpurun_out/${T}_bench_samebox.err
fi
# the reference-shaped three-call sequence (sample! / solve! / llikelihood as separate launches), same box
python scripts/gpu_ext_probe.py 2>/dev/null | grep "B/path-step" > gpurun_out/${T}_separate_calls.txt
