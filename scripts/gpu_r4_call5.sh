#!/bin/bash
# Round 4, fifth GPU call: per-channel (max / min over the 128 L2 channels) counters of contiguous (always slow) against plain allocations
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4e; mkdir -p $O
export TMPDIR=/tmp
cd $R
export PMC_NENS=6 PMC_PLACES="contig contig malloc malloc malloc malloc"
declare -A SETS
SETS[a]="BH_TCC_REQ_max BH_TCC_REQ_min BH_TCC_REQ_sum BH_TCC_BUSY_max BH_TCC_BUSY_min BH_TCC_BUSY_sum BH_TCC_EA0_WRREQ_max BH_TCC_EA0_WRREQ_min BH_TCC_TAG_STALL_max BH_TCC_TAG_STALL_min BH_TCC_TAG_STALL_sum"
SETS[b]="BH_TCC_WRITE_max BH_TCC_WRITE_min BH_TCC_EA0_WRREQ_STALL_max BH_TCC_EA0_WRREQ_STALL_min BH_TCC_EA0_WRREQ_STALL_sum BH_TCC_EA0_RDREQ_max BH_TCC_EA0_RDREQ_min BH_TCC_TOO_MANY_EA_WRREQS_STALL_max BH_TCC_TOO_MANY_EA_WRREQS_STALL_sum"
for s in a b; do
  (cd /tmp && rm -rf /tmp/pmc_$s && timeout 600 rocprofv3 -E $R/scripts/r4_extra_counters.yaml --kernel-trace --pmc ${SETS[$s]} -d /tmp/pmc_$s -o t -- python $R/scripts/gpu_place_pmc.py) > $O/pmc_$s.log 2>&1
  f=$(ls /tmp/pmc_$s/*.db 2>/dev/null | head -1)
  grep "^ensemble" $O/pmc_$s.log > $O/pmc_$s.txt
  [ -n "$f" ] && python scripts/rocpd_dispatches.py $f k_pc 7 >> $O/pmc_$s.txt 2>&1
  grep -E "^ensemble|^blk|^ +#" $O/pmc_$s.txt | cut -c1-400
  tail -5 $O/pmc_$s.log | cut -c1-300
done
# the headline's live traffic pass inside bench.py, and the large-gap placement of W against Xo inside ONE contiguous block
timeout 900 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); r=d['roofline']; print({k: r.get(k) for k in ('kernel_avg_ms','frac','traffic','traffic_box','traffic_over_algorithmic','traffic_live_failed','traffic_source')})"
PROBE_GAPS="0 1048576 4194304 8388608 16777216 33554432 67108864" timeout 600 python scripts/gpu_delta_probe.py 2>&1 | grep -v amdgpu.ids > $O/delta_far.txt; cat $O/delta_far.txt
