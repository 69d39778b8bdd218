#!/bin/bash
# Round 4, fourth GPU call: leading-dimension sweep under contiguous / plain allocations; issue priorities of the small kernels
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4d; mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 1500 python scripts/gpu_ldpad_probe.py 2>&1 | grep -v amdgpu.ids > $O/ldpad_probe.txt; cat $O/ldpad_probe.txt
timeout 600 python scripts/gpu_tune_probe.py 2>&1 | grep -v amdgpu.ids > $O/tune_probe.txt; cat $O/tune_probe.txt
