mkdir -p gpurun_out/r4x
python -m pytest tests -x -q -m gpu 2>&1 | tail -12 > gpurun_out/r4x/tests1.txt
python -m pytest tests -x -q -m gpu 2>&1 | tail -12 > gpurun_out/r4x/tests2.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r4x/smoke.txt 2>&1
