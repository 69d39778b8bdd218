mkdir -p gpurun_out/r4ag
for w in 128 192 256 320 384 512 768 256; do echo "statistics-pass workgroups $w (K = 4, L = 4)"; BHIP_SEG_MCNEXT_WGS=$w python scripts/gpu_smooth_ab.py one "ring"; done 2>&1 | grep -v amdgpu.ids > gpurun_out/r4ag/wgs.txt
