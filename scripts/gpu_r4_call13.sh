#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4m; mkdir -p $O
export TMPDIR=/tmp
cd $R
PROBE_DIMS="8 9 10 11 12" timeout 900 python scripts/gpu_mid_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/mid_probe.txt
PROBE_DIMS="8 9 10 12" timeout 900 python scripts/gpu_midchain_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/midchain_probe.txt
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; grep -E "passed|failed|rror" $O/pytest.txt | tail -3
