mkdir -p gpurun_out/r4v
python -m pytest tests/test_gpu_tile.py tests/test_gpu_usertile.py -x -q -m gpu 2>&1 | tail -4 > gpurun_out/r4v/tests.txt
PROBE_DIMS="4 8 9 10 11 12 14" python scripts/gpu_mid_probe.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r4v/mid_probe.txt
PROBE_DIMS="8 9 10 12" python scripts/gpu_midchain_probe.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r4v/midchain_probe.txt
