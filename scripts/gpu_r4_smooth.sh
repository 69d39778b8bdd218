mkdir -p gpurun_out/r4ao
for i in 1 2 3; do python bench.py --no-other-modes --no-cpu-baseline --no-live-traffic > gpurun_out/r4ao/b$i.json 2> gpurun_out/r4ao/b$i.err; done
