mkdir -p gpurun_out/r4ak
for r in 1 2; do ./ab/alloc_sequence_probe 28 4; done > gpurun_out/r4ak/seq.txt 2>&1
./ab/alloc_sequence_probe 28 4 plain >> gpurun_out/r4ak/seq.txt 2>&1
./ab/alloc_sequence_probe 12 16 >> gpurun_out/r4ak/seq.txt 2>&1
