mkdir -p gpurun_out/r4af
python -m pytest tests/test_gpu_segchains.py -x -q -m gpu 2>&1 | tail -25 > gpurun_out/r4af/tests.txt
