"""The oracle under AddressSanitizer + UndefinedBehaviorSanitizer (auxiliary subsystems: sanitizers).  The checker is what every
parity claim rests on: its own pins (tests/test_oracle.py: golden vectors, K1-K14, the noise specifications) and the independent numpy
restatement run once more against a build of the SAME source with -fsanitize=address,undefined (`make -C oracle san`), in a child
process with the sanitizer runtimes preloaded; any out-of-bounds access, use after free, signed overflow, misaligned or null access in
oracle/bridge_oracle.c aborts the child (-fno-sanitize-recover)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _runtime(name):
    try:
        p = subprocess.run(["gcc", f"-print-file-name={name}"], capture_output=True, text=True, timeout=30).stdout.strip()
    except Exception:
        return None
    return p if os.path.isabs(p) and os.path.exists(p) else None


def test_oracle_pins_pass_under_asan_and_ubsan():
    asan, ubsan = _runtime("libasan.so"), _runtime("libubsan.so")
    if not asan or not ubsan:
        pytest.skip("no sanitizer runtimes in this image")
    so = os.path.join(ROOT, "oracle", "libbridge_oracle_san.so")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "san"])
    env = dict(os.environ, LD_PRELOAD=f"{asan}:{ubsan}", ASAN_OPTIONS="detect_leaks=0:verify_asan_link_order=0", UBSAN_OPTIONS="print_stacktrace=1",
               BRIDGE_ORACLE_SO=so, OMP_NUM_THREADS="2")
    code = ("import sys, os; sys.path.insert(0, %r); import oracle as o; assert o._SO.endswith('_san.so'); o.load(); "
            "print('SANITIZED', [l.split()[-1] for l in open('/proc/self/maps') if 'libbridge_oracle' in l][0])") % os.path.join(ROOT, "tests")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0 and "SANITIZED" in r.stdout and "libbridge_oracle_san.so" in r.stdout, r.stdout + r.stderr
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", os.path.join(ROOT, "tests", "test_oracle.py"),
                        os.path.join(ROOT, "tests", "test_independent_restatement.py")], capture_output=True, text=True, env=env, timeout=1500, cwd=ROOT)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-3000:]
    assert "runtime error" not in out and "AddressSanitizer" not in out, out[-3000:]
    assert " passed" in out
