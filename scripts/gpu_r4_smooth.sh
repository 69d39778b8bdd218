mkdir -p gpurun_out/r4s
for w in 128 192 256 320 384 256; do echo "two streams, mcnext workgroups $w"; BHIP_SEG_MCNEXT_WGS=$w python scripts/gpu_smooth_ab.py one "tb"; done 2>&1 | grep -v amdgpu.ids > gpurun_out/r4s/wgs.txt
BHIP_SEG_PLAIN_X=1 python scripts/gpu_smooth_ab.py one "plain two streams" 2>&1 | grep -v amdgpu.ids >> gpurun_out/r4s/wgs.txt
BHIP_SEG_PLAIN_X=1 BHIP_SEG_ONE_STREAM=1 python scripts/gpu_smooth_ab.py one "plain one stream" 2>&1 | grep -v amdgpu.ids >> gpurun_out/r4s/wgs.txt
