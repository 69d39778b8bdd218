mkdir -p gpurun_out/r4z
python -m pytest tests -x -q -m gpu 2>&1 | tail -12 > gpurun_out/r4z/tests.txt
T=r4 bash scripts/gpu_smooth_profile.sh > gpurun_out/r4z/prof.log 2>&1
python scripts/gpu_smooth_ab.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r4z/ab.txt
python bench.py > gpurun_out/r4z/bench.json 2> gpurun_out/r4z/bench.err
