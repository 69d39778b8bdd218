mkdir -p gpurun_out/r4t
python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r4t/tests.txt
T=r4 bash scripts/gpu_smooth_profile.sh > gpurun_out/r4t/prof.log 2>&1
python scripts/gpu_smooth_ab.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r4t/ab.txt
python bench.py > gpurun_out/r4t/bench.json 2> gpurun_out/r4t/bench.err
