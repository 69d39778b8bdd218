#!/bin/bash
# ONE parametrised runner for the GPU box (replaces the one-shot gpu_r4_call*.sh wrappers of round 4):
#     gpurun --timeout 1500 -- 'bash scripts/gpu_call.sh <tag> <step> [<step> ...]'
# Every step writes under gpurun_out/<tag>/ (merged back by gpurun); nothing else persists.  Steps:
#     tests            the whole GPU suite (pytest -m gpu), tail to tests.txt
#     testsall         the same without -x (every failure listed)
#     tests:<expr>     pytest -m gpu -k <expr>
#     smoke            __graft_entry__.smoke()
#     bench            the default bench line (fresh process)            -> bench.json / bench.err
#     bench2           once more (second fresh process)                   -> bench2.json
#     quick            bench.py --no-cpu-baseline --no-live-traffic       -> quick.json   (headline + the "modes" object, ~1 min)
#     mode:<m>[:P]     bench.py --mode <m> [--chains P] --no-other-modes --no-cpu-baseline --steps 200   -> mode_<m>[_P].json
#     rdf              tests/rng_device_forms.hip (exhaustive device checks of the generator)
#     env:<K>=<V>      export an environment variable for the following steps (e.g. env:BHIP_PC_LARGE_NPAIR=1)
#     py:<script> [args]  python <script> [args] (a probe under scripts/), stdout to <basename>.txt
#     profile:<m>      scripts/gpu_profile.sh <tag>_<m> with the warm protocol (see that script)
set -o pipefail   # a step's rc is its command's, not that of the `tail` behind it (a failing suite read "tests rc=0" until the end of round 5)
TAG=${1:?tag}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
for step in "$@"; do
  case $step in
    tests) timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -25 > $OUT/tests.txt ;;
    testsall) timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -80 > $OUT/testsall.txt ;;
    tests:*) timeout 1500 python -m pytest tests -x -q -m gpu -k "${step#tests:}" 2>&1 | tail -40 > $OUT/tests_k.txt ;;
    smoke) timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1 ;;
    bench) timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err ;;
    bench2) timeout 900 python bench.py > $OUT/bench2.json 2> $OUT/bench2.err ;;
    quick) timeout 900 python bench.py --no-cpu-baseline --no-live-traffic > $OUT/quick.json 2> $OUT/quick.err ;;
    mode:*) IFS=: read -r _ m p <<< "$step"
            timeout 600 python bench.py --mode $m ${p:+--chains $p} --no-other-modes --no-cpu-baseline --no-live-traffic --steps 200 --warmup 5 \
              > $OUT/mode_${m}${p:+_$p}${SUFFIX}.json 2> $OUT/mode_${m}${p:+_$p}${SUFFIX}.err ;;
    rdf) /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -ffp-contract=off -I bridge.jl_amd/csrc tests/rng_device_forms.hip -o /tmp/rdf 2> /dev/null \
           && timeout 300 /tmp/rdf > $OUT/rdf.txt 2>&1 ;;
    env:*) kv=${step#env:}; export "$kv"; SUFFIX="_${kv//[^A-Za-z0-9]/}" ;;
    py:*) s=${step#py:}; n=$(basename ${s%% *}); timeout 900 python $s > $OUT/${n%.py}.txt 2>&1 ;;   # ("py:script.py arg ...": arguments allowed)
    profile:*) bash scripts/gpu_profile.sh ${TAG}_${step#profile:} --mode ${step#profile:} ;;
    *) echo "unknown step $step" >> $OUT/errors.txt ;;
  esac
  echo "$step rc=$?" >> $OUT/steps.txt
done
tail -n 30 $OUT/steps.txt
