T=r4 bash scripts/gpu_profile_all.sh > gpurun_out/r4_profile_all.log 2>&1
