#!/usr/bin/env python3
"""Per-kernel resources as the CODE OBJECT states them (what the hardware allocates by): .vgpr_count, .agpr_count,
.sgpr_count, LDS (.group_segment_fixed_size; dynamic LDS is added at launch), scratch (.private_segment_fixed_size),
and the waves per SIMD the register count allows (512-entry file, granule 8; MI355X_MICROARCH.md "Register files").

    python scripts/codeobj_resources.py [substring ...]  > profiles/r2_codeobj_resources.txt

Reads the gfx950 code objects embedded in bridge.jl_amd/csrc/build/*.o (llvm-objcopy + clang-offload-bundler + llvm-readelf)."""
import glob
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernels(obj):
    with tempfile.TemporaryDirectory() as td:
        fat, co = os.path.join(td, "fat.bin"), os.path.join(td, "k.co")
        # (with an explicit output file: llvm-objcopy otherwise rewrites `obj` IN PLACE -- a fresh mtime on a stale object makes `make` skip it)
        subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", obj, os.path.join(td, "copy.o")], check=True, stderr=subprocess.DEVNULL)
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True, stderr=subprocess.DEVNULL)
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    out = []
    for blk in re.split(r"\n\s*- \.agpr_count:", notes)[1:]:
        blk = ".agpr_count:" + blk
        g = lambda key: (re.search(rf"\.{key}:\s+(\S+)", blk) or [None, "?"])[1]
        name = g("name")
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        out.append(dict(name=dem, vgpr=g("vgpr_count"), agpr=g("agpr_count"), sgpr=g("sgpr_count"), lds=g("group_segment_fixed_size"),
                        scratch=g("private_segment_fixed_size"), wg=g("max_flat_workgroup_size")))
    return out


def waves(v, a):
    try:
        tot = (int(v) + int(a) + 7) // 8 * 8
        return min(8, 512 // max(tot, 8))
    except ValueError:
        return "?"


def main():
    pats = sys.argv[1:] or ["k_pc<", "k_paths<", "k_chain_lines<", "k_tile<", "k_wiener", "k_girsanov"]
    print(f"{'kernel':<84} {'vgpr':>5} {'agpr':>5} {'sgpr':>5} {'waves/SIMD':>10} {'static LDS':>10} {'scratch':>8} {'max wg':>7}")
    for obj in sorted(glob.glob(os.path.join(ROOT, "bridge.jl_amd", "csrc", "build", "*.o"))):
        for k in kernels(obj):
            if any(p in k["name"] for p in pats):
                print(f"{k['name'][:84]:<84} {k['vgpr']:>5} {k['agpr']:>5} {k['sgpr']:>5} {waves(k['vgpr'], k['agpr']):>10} {k['lds']:>10} {k['scratch']:>8} {k['wg']:>7}")


if __name__ == "__main__":
    main()
