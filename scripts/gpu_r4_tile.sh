mkdir -p gpurun_out/r4ae
BENCH_SAME_DEVICE=1 python bench.py --gpus 8 --steps 10 --warmup 3 > gpurun_out/r4ae/bench8.json 2> gpurun_out/r4ae/bench8.err
BENCH_SAME_DEVICE=1 python bench.py --gpus 2 --steps 10 --warmup 3 > gpurun_out/r4ae/bench2.json 2> gpurun_out/r4ae/bench2.err
gcc -O2 -o /tmp/fhn_multi examples/fhn_chains_multi.c -Iinclude -Lbridge.jl_amd -lbridgehip -lm -Wl,-rpath,$PWD/bridge.jl_amd && /tmp/fhn_multi > gpurun_out/r4ae/c_example.txt 2>&1
