#!/bin/bash
# Round 4, sixth GPU call: where W and Xo have to lie (explicit offsets in one contiguous block); per-channel counters contiguous vs plain
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4f; mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 1500 python scripts/gpu_far_probe.py 2>&1 | grep -v amdgpu.ids > $O/far_probe.txt; cat $O/far_probe.txt
export PMC_NENS=6 PMC_PLACES="contig contig2:0:40960 malloc malloc malloc malloc"
declare -A SETS
SETS[a]="BH_TCC_REQ_max BH_TCC_REQ_min BH_TCC_REQ_sum BH_TCC_BUSY_max BH_TCC_BUSY_min BH_TCC_BUSY_sum BH_TCC_EA0_WRREQ_max BH_TCC_EA0_WRREQ_min BH_TCC_TAG_STALL_max BH_TCC_TAG_STALL_min BH_TCC_TAG_STALL_sum"
SETS[b]="BH_TCC_WRITE_max BH_TCC_WRITE_min BH_TCC_EA0_WRREQ_STALL_max BH_TCC_EA0_WRREQ_STALL_min BH_TCC_EA0_WRREQ_STALL_sum BH_TCC_EA0_RDREQ_max BH_TCC_EA0_RDREQ_min BH_TCC_TOO_MANY_EA_WRREQS_STALL_max BH_TCC_TOO_MANY_EA_WRREQS_STALL_sum"
for s in a b; do
  (cd /tmp && rm -rf /tmp/pmc_$s && timeout 600 rocprofv3 -E $R/scripts/r4_extra_counters.yaml --kernel-trace --pmc ${SETS[$s]} -d /tmp/pmc_$s -o t -- python $R/scripts/gpu_place_pmc.py) > $O/pmc_$s.log 2>&1
  f=$(ls /tmp/pmc_$s/*.db 2>/dev/null | head -1)
  grep "^ensemble" $O/pmc_$s.log > $O/pmc_$s.txt
  [ -n "$f" ] && python scripts/rocpd_dispatches.py $f k_pc 7 >> $O/pmc_$s.txt 2>&1
  grep -E "^ensemble|^blk|^ +#" $O/pmc_$s.txt | cut -c1-400
  tail -3 $O/pmc_$s.log | cut -c1-200
done
