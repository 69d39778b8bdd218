#!/bin/bash
# Profiling recipe (run on the GPU box through gpurun):  bash scripts/gpu_profile.sh <tag> --mode <mode> [more bench args]
#   1. rocprofv3 --kernel-trace --stats of `bench.py --mode <mode>` UNDER THE BENCH'S OWN PROTOCOL -- warm-up, the 300 queued pre-warm steps,
#      then 300 timed launches -- so that the trace and the HIP events of the SAME call can be compared launch for launch
#      (scripts/profile_check.py: avg / median / min of the traced launches, of the 300 timed ones, the HIP-event figure of the JSON line,
#      the shader clock rocm-smi reported while the kernel ran; fails when median and HIP events differ by more than 5 %);
#   2. separate --pmc passes (never combined with other trace domains; 20 steps: counters do not depend on the clocks): FETCH_SIZE,
#      WRITE_SIZE, two SQ sets.
# Text summaries land in gpurun_out/<tag>_*.txt (copy the ones to be judged into profiles/).  rc = 1 when the check of step 1 fails.
set -o pipefail   # a step's exit code is its command's, not that of the `tail` / `tee` behind it
TAG=${1:-prof}; shift
ARGS=${@:---mode mcmc}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
TSTEPS=${TSTEPS:-300}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
COMMON="--no-cpu-baseline --no-other-modes --no-live-traffic"
# the shader clock while the traced run is busy: rocm-smi once a second beside it, the largest sclk seen
( while true; do rocm-smi --showclocks 2>/dev/null | grep -i "sclk" | grep -o "([0-9]*Mhz)" | tr -d "()Mhz" ; sleep 1; done > $OUT/sclk.txt ) &
SMI=$!
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python $R/bench.py $ARGS --steps $TSTEPS --warmup 5 $COMMON > $OUT/trace.log 2>&1
kill $SMI 2>/dev/null; wait $SMI 2>/dev/null
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o t -- python $R/bench.py $ARGS --steps 20 --warmup 2 $COMMON > $OUT/fetch.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o t -- python $R/bench.py $ARGS --steps 20 --warmup 2 $COMMON > $OUT/write.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $OUT/sq -o t -- python $R/bench.py $ARGS --steps 20 --warmup 2 $COMMON > $OUT/sq.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $OUT/sq2 -o t -- python $R/bench.py $ARGS --steps 20 --warmup 2 $COMMON > $OUT/sq2.log 2>&1
rc=0
for d in trace fetch write sq sq2; do
  f=$(ls $OUT/$d/*.db 2>/dev/null | head -1)
  [ -n "$f" ] && python $R/scripts/rocpd_summary.py $f > $R/gpurun_out/${TAG}_$d.txt 2>&1
  grep -h '"metric"' $OUT/$d.log | tail -1 >> $R/gpurun_out/${TAG}_$d.txt
  if [ $d = trace ] && [ -n "$f" ]; then
    python $R/scripts/profile_check.py $f $OUT/trace.log $OUT/sclk.txt $TSTEPS > $R/gpurun_out/${TAG}_check.txt 2>&1 || rc=1
    cat $R/gpurun_out/${TAG}_check.txt
  fi
  rm -rf $OUT/$d
done
exit $rc
