#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4g; mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 1800 python scripts/gpu_arena_probe.py 2>&1 | grep -v amdgpu.ids > $O/arena_probe.txt; cat $O/arena_probe.txt
