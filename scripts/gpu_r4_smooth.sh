mkdir -p gpurun_out/r4ai
python -m pytest tests/test_gpu_adapt_device.py tests/test_gpu_segchains.py tests/test_c_example.py tests/test_linearappr.py -x -q -m gpu 2>&1 | tail -12 > gpurun_out/r4ai/tests.txt
python scripts/gpu_smooth_ab.py one "ring" 2>&1 | grep -v amdgpu.ids > gpurun_out/r4ai/ab.txt
BHIP_SEG_ONE_STREAM=1 python scripts/gpu_smooth_ab.py one "ring one stream" 2>&1 | grep -v amdgpu.ids >> gpurun_out/r4ai/ab.txt
python scripts/gpu_smooth_ab.py one "ring" 2>&1 | grep -v amdgpu.ids >> gpurun_out/r4ai/ab.txt
