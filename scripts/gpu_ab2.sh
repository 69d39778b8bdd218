#!/bin/bash
# A/B on one box: the default library and every bridge.jl_amd/variants/*.so through scripts/gpu_small_probe.py
#   PROBE_SIZES=32768,65536 bash scripts/gpu_ab2.sh
set -o pipefail   # a step's exit code is its command's, not that of the `tail` / `tee` behind it
for so in default $(ls bridge.jl_amd/variants/*.so 2>/dev/null); do
  if [ $so = default ]; then unset BRIDGEHIP_SO; else export BRIDGEHIP_SO=$PWD/$so; fi
  echo "=== $so"
  timeout 600 python scripts/gpu_small_probe.py 2>&1 | grep -v amdgpu.ids
done
