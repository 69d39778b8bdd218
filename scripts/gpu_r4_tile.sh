R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4ah; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_BUSY_CYCLES -d $OUT/lds -o t -- python $R/bench.py --mode c2 --steps 8 --warmup 2 --no-cpu-baseline --no-other-modes > $OUT/lds.log 2>&1
f=$(ls $OUT/lds/*.db 2>/dev/null | head -1); [ -n "$f" ] && python $R/scripts/rocpd_summary.py $f > $R/gpurun_out/r4ah/c2_lds.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_BUSY_CYCLES -d $OUT/lds2 -o t -- python $R/bench.py --mode proposals --steps 8 --warmup 2 --no-cpu-baseline --no-other-modes > $OUT/lds2.log 2>&1
f=$(ls $OUT/lds2/*.db 2>/dev/null | head -1); [ -n "$f" ] && python $R/scripts/rocpd_summary.py $f > $R/gpurun_out/r4ah/proposals_lds.txt 2>&1
rm -rf $OUT/lds $OUT/lds2
