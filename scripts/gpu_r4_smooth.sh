mkdir -p gpurun_out/r4al
for i in 1 2 3 4 5 6; do python bench.py --no-other-modes --no-cpu-baseline --no-live-traffic > gpurun_out/r4al/b$i.json 2> gpurun_out/r4al/b$i.err; done
python -m pytest tests/test_gpu_pc.py tests/test_gpu_segchains.py tests/test_gpu_group.py -x -q -m gpu 2>&1 | tail -12 > gpurun_out/r4al/tests.txt
