"""bench.py looks the PMC traffic of a mode up by the EXACT name of the kernel the current build launches for it
(profiles/r3_<mode>_{fetch,write}.txt).  Kernel templates gain arguments now and then; this test keeps the name builders
honest against the code objects of the current build (CPU only: reads bridge.jl_amd/csrc/build/*.o with the LLVM tools),
and checks that the committed profile summaries carry those names (no STALE / MISSING traffic in the bench line)."""
import glob
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))


def test_bench_kernel_names_exist_in_the_build_and_in_the_profiles():
    objs = glob.glob(os.path.join(ROOT, "bridge.jl_amd", "csrc", "build", "*.o"))
    if not objs or not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-readelf") or shutil.which("c++filt") is None:
        pytest.skip("no build directory / LLVM tools here")
    import bench
    import codeobj_resources as cr
    built = set()
    for o in objs:
        built.update(k["name"] for k in cr.kernels(o))
    for mode, spec in bench.MODES.items():
        for P in {spec[4], 32768, 65536, 262144}:
            name = spec[8](P)
            assert any(name + "(" in b for b in built), (mode, P, name)
        tr, src = bench.profiled_traffic(mode, spec[8](spec[4]))
        assert tr is not None and tr > 0, (mode, src)
