#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4h; mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 1800 python scripts/gpu_spacer_probe.py 2>&1 | grep -v amdgpu.ids > $O/spacer_probe.txt; cat $O/spacer_probe.txt
for i in 1 2 3; do BHIP_PLACE=spacer:96 PROBE_SPECS="spacer:96" timeout 300 python scripts/gpu_spacer_probe.py 2>&1 | grep -v amdgpu.ids | tee -a $O/spacer_fresh.txt; done
PROBE_MODES="nclar_mcmc linpro32_mcmc" PROBE_SPECS="spacer:96 malloc" timeout 900 python scripts/gpu_spacer_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/spacer_other.txt
