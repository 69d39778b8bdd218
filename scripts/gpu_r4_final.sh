# end-of-round check: the whole GPU suite, smoke, the default bench line (fresh process)
mkdir -p gpurun_out/r4fin
python -m pytest tests -x -q -m gpu 2>&1 | tail -12 > gpurun_out/r4fin/tests.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4fin/smoke.txt 2>&1
python bench.py > gpurun_out/r4fin/bench.json 2> gpurun_out/r4fin/bench.err
python bench.py > gpurun_out/r4fin/bench2.json 2> gpurun_out/r4fin/bench2.err
