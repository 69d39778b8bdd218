#!/bin/bash
# Profiling recipe (run on the GPU box through gpurun):  bash scripts/gpu_profile.sh <tag> [bench args]
#   1. rocprofv3 --kernel-trace --stats of the bench command  -> per-kernel durations / registers
#   2. separate --pmc passes (never combined with other trace domains): FETCH_SIZE, WRITE_SIZE, SQ mix
# Text summaries land in gpurun_out/<tag>_*.txt (copy the ones to be judged into profiles/).
TAG=${1:-prof}; shift
ARGS=${@:---steps 6 --warmup 2 --no-cpu-baseline --no-other-modes}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python $R/bench.py $ARGS > $OUT/trace.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o t -- python $R/bench.py $ARGS > $OUT/fetch.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o t -- python $R/bench.py $ARGS > $OUT/write.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $OUT/sq -o t -- python $R/bench.py $ARGS > $OUT/sq.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $OUT/sq2 -o t -- python $R/bench.py $ARGS > $OUT/sq2.log 2>&1
for d in trace fetch write sq sq2; do
  f=$(ls $OUT/$d/*.db 2>/dev/null | head -1)
  [ -n "$f" ] && python $R/scripts/rocpd_summary.py $f > $R/gpurun_out/${TAG}_$d.txt 2>&1
  grep -h '"metric"' $OUT/$d.log | tail -1 >> $R/gpurun_out/${TAG}_$d.txt
  rm -rf $OUT/$d
done
head -12 $R/gpurun_out/${TAG}_trace.txt
