mkdir -p gpurun_out/r4ac
python scripts/gpu_smooth_ab.py one "ring" 2>&1 | grep -v amdgpu.ids > gpurun_out/r4ac/ab.txt
python scripts/gpu_smooth_ab.py one "ring" 2>&1 | grep -v amdgpu.ids >> gpurun_out/r4ac/ab.txt
python -m pytest tests/test_gpu_segchains.py tests/test_gpu_adapt_device.py -x -q -m gpu 2>&1 | tail -12 > gpurun_out/r4ac/tests.txt
