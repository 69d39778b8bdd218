#!/bin/bash
# Round 4, second GPU call: the GPU suite (noise specifications, group stepping), the XCD-mapping probe, the default bench line
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4b; mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; grep -E "passed|failed|error" $O/pytest.txt | tail -3
timeout 900 python scripts/gpu_xcd_probe.py 2>&1 | grep -v amdgpu.ids > $O/xcd_probe.txt; cat $O/xcd_probe.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 2200 $O/bench.json
