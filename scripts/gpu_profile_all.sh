#!/bin/bash
# Profile set of a round (T=r5 by default): every bench mode through scripts/gpu_profile.sh -- kernel trace under the bench's own warm
# protocol (300 queued steps, 300 traced + HIP-event-timed launches, scripts/profile_check.py) + separate FETCH / WRITE / SQ passes --, then
# the default bench line ON THE SAME BOX.  Summaries land in gpurun_out/${T}_<mode>_{trace,fetch,write,sq,sq2,check}.txt; copy them into
# profiles/.  rc = 1 when any mode's trace and HIP events differ by more than 5 %.
set -o pipefail   # a step's exit code is its command's, not that of the `tail` / `tee` behind it
T=${T:-r5}
rc=0
for m in ${MODES:-mcmc c4shard c2 proposals nclar nclar_mcmc linpro4 linpro32 linpro32_mcmc}; do
  case $m in linpro32*) export TSTEPS=60;; *) export TSTEPS=300;; esac
  bash scripts/gpu_profile.sh ${T}_$m --mode $m > gpurun_out/${T}_${m}_profile.log 2>&1 || { rc=1; echo "CHECK FAILED: $m"; }
  tail -6 gpurun_out/${T}_${m}_check.txt
done
if [ -z "$NO_BENCH" ]; then
  mkdir -p profiles && cp gpurun_out/${T}_*_{trace,fetch,write,sq,sq2,check}.txt profiles/ 2>/dev/null
  python bench.py > gpurun_out/${T}_bench_samebox.json 2> gpurun_out/${T}_bench_samebox.err
fi
# the reference-shaped three-call sequence (sample! / solve! / llikelihood as separate launches), same box
python scripts/gpu_ext_probe.py 2>/dev/null | grep "B/path-step" > gpurun_out/${T}_separate_calls.txt
echo "profile set rc=$rc"
exit $rc
