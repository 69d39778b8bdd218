#!/bin/bash
# headline kernel vs the gap between W and Xo inside their common allocation; 3 fresh allocations per gap (same process)
for gap in "$@"; do
  echo "== gap $gap KB"
  BHIP_XO_GAP_KB=$gap PROBE_REPS=3 python scripts/gpu_alloc_probe.py 2>&1 | grep -v amdgpu.ids
done
