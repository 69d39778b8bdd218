mkdir -p gpurun_out/r4s
python -m pytest tests/test_gpu_segchains.py tests/test_gpu_adapt_device.py tests/test_c_example.py tests/test_linearappr.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r4s/tests.txt
python scripts/gpu_smooth_ab.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r4s/ab.txt
