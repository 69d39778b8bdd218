#!/bin/bash
# Profile an arbitrary command (run on the GPU box through gpurun):  bash scripts/gpu_prof_cmd.sh <tag> <passes> -- <cmd...>
#   passes: comma list of  trace,fetch,write,sq,sq2   (each its own run; PMC passes never combined with other trace domains)
# Text summaries land in gpurun_out/<tag>_<pass>.txt (copy the ones to be judged into profiles/).
set -o pipefail   # a step's exit code is its command's, not that of the `tail` / `tee` behind it
TAG=$1; PASSES=$2; shift 3
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
declare -A PMC
PMC[fetch]="FETCH_SIZE"
PMC[write]="WRITE_SIZE"
PMC[sq]="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
PMC[sq2]="SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
for p in ${PASSES//,/ }; do
  if [ $p = trace ]; then
    (cd $R && rocprofv3 --kernel-trace --stats -d $OUT/$p -o t -- "$@") > $OUT/$p.log 2>&1
  else
    (cd $R && rocprofv3 --kernel-trace --pmc ${PMC[$p]} -d $OUT/$p -o t -- "$@") > $OUT/$p.log 2>&1
  fi
  f=$(ls $OUT/$p/*.db 2>/dev/null | head -1)
  [ -n "$f" ] && python $R/scripts/rocpd_summary.py $f > $R/gpurun_out/${TAG}_$p.txt 2>&1
  grep -h -v "amdgpu.ids\|^W2\|^E2\|rocprofv3" $OUT/$p.log | tail -15 >> $R/gpurun_out/${TAG}_$p.txt
  rm -rf $OUT/$p
done
