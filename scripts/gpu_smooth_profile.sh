#!/bin/bash
# the smoothing loop (bench.py's `smoothing` record) under rocprofv3: kernel trace, then FETCH_SIZE and WRITE_SIZE in their own passes
set -o pipefail   # a step's exit code is its command's, not that of the `tail` / `tee` behind it
R=$GRAFT_REPO_ROOT; T=${T:-r4}; OUT=$R/gpurun_out/${T}_smooth; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python $R/scripts/gpu_smooth_probe.py > $OUT/trace.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o t -- python $R/scripts/gpu_smooth_probe.py > $OUT/fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o t -- python $R/scripts/gpu_smooth_probe.py > $OUT/write.log 2>&1
for d in trace fetch write; do
  f=$(ls $OUT/$d/*.db 2>/dev/null | head -1)
  [ -n "$f" ] && python $R/scripts/rocpd_summary.py $f > $R/gpurun_out/${T}_smoothing_$d.txt 2>&1
  tail -3 $OUT/$d.log >> $R/gpurun_out/${T}_smoothing_$d.txt
  rm -rf $OUT/$d
done
