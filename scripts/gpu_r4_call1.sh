#!/bin/bash
# Round 4, first GPU call: (1) the GPU suite on the arena refactor; (2) placement by construction: hipMalloc vs the virtual-memory
# API (scripts/gpu_place_probe.py); (3) s_memtime budgets of the wave-specialised kernels (scripts/gpu_stamp_probe.py);
# (4) memory-side counters of slow vs fast allocations in ONE process, per dispatch (scripts/gpu_place_pmc.py + rocpd_dispatches.py)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4a; mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
(cd /tmp && rocprofv3 -L > $O/counters_avail.txt 2>&1)
timeout 900 python scripts/gpu_place_probe.py > $O/place_probe.txt 2>&1; tail -5 $O/place_probe.txt
BRIDGEHIP_SO=$R/ab/stamp1.so timeout 600 python scripts/gpu_stamp_probe.py > $O/stamp1.txt 2>&1
BRIDGEHIP_SO=$R/ab/stamp2.so STAMP_MODES="c2 c4shard proposals64k" timeout 600 python scripts/gpu_stamp_probe.py > $O/stamp2.txt 2>&1
cat $O/stamp1.txt $O/stamp2.txt | grep -v amdgpu.ids
declare -A SETS
SETS[wr]="TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_64B_sum TCC_TAG_STALL_sum"
SETS[rd]="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_GMI_CREDIT_STALL_sum"
SETS[lvl]="TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_IO_CREDIT_STALL_sum"
SETS[tlb]="TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_PENDING_STALL_CYCLES_sum"
SETS[chan]="TCC_EA0_WRREQ TCC_EA0_RDREQ"
SETS[chan2]="TCC_EA0_WRREQ_STALL TCC_TAG_STALL"
SETS[sq]="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
for s in wr rd lvl tlb chan chan2 sq; do
  want=""
  for c in ${SETS[$s]}; do
    if grep -qw "$c" $O/counters_avail.txt; then want="$want $c"; else echo "counter $c: not in rocprofv3 -L" >> $O/pmc_$s.txt; fi
  done
  [ -z "$want" ] && continue
  (cd /tmp && rm -rf /tmp/pmc_$s && timeout 600 rocprofv3 --kernel-trace --pmc $want -d /tmp/pmc_$s -o t -- python $R/scripts/gpu_place_pmc.py) > $O/pmc_$s.log 2>&1
  f=$(ls /tmp/pmc_$s/*.db 2>/dev/null | head -1)
  grep "^ensemble" $O/pmc_$s.log >> $O/pmc_$s.txt
  [ -n "$f" ] && python scripts/rocpd_dispatches.py $f k_pc 7 >> $O/pmc_$s.txt 2>&1
  [ $s = wr ] && [ -n "$f" ] && cp $f $O/pmc_wr.db
  tail -8 $O/pmc_$s.txt | cut -c1-220
done
