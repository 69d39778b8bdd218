#!/usr/bin/env python
"""Does the committed rocprofv3 trace reproduce the bench line of the SAME call?  

    python scripts/profile_check.py <trace.db> <bench stdout/stderr log> <sclk samples> <timed steps>

Reads the dominant kernel of the run from the bench JSON line (roofline.kernel, kernel_avg_ms = HIP events over the K timed steps), takes that
kernel's dispatches from the rocpd database in launch order and prints, in microseconds: avg / median / min over ALL traced launches
(warm-up, set-up, the 300 queued pre-warm steps, the timed steps) and over the LAST K (the timed region itself), the HIP-event average, the
ratio median(last K) / HIP events and the largest shader clock rocm-smi reported during the run.  Exit code 1 when the two differ by more
than 5 %: the fraction a reader derives from profiles/ must be the one the bench prints."""
import json
import sqlite3
import sys

import numpy as np


def main():
    db, log, sclk, K = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
    line = [l for l in open(log, errors="replace").read().splitlines() if l.startswith("{") and '"metric"' in l]
    if not line:
        print("profile_check: no bench JSON line in", log)
        return 1
    j = json.loads(line[-1])
    kern, hip_ms, mode = j["roofline"]["kernel"], j["roofline"]["kernel_avg_ms"], j["config"]["mode"]
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute("select name, start, duration from kernels order by start"))
    d = np.array([r[2] for r in rows if kern in r[0]], dtype=np.float64) / 1e3
    if len(d) < K:
        print(f"profile_check: only {len(d)} launches of {kern} in the trace, {K} timed steps expected")
        return 1
    # (since the end of round 5 the bench follows its K timed launches -- back to back between two events -- with an UNTIMED pass of K more
    # that carry an event between consecutive launches: roofline.per_launch_pass; the timed region is then the K launches before those)
    tail = K if j["roofline"].get("per_launch_pass") else 0
    if len(d) < K + tail:
        print(f"profile_check: only {len(d)} launches of {kern} in the trace, {K + tail} expected")
        return 1
    last = d[-K - tail:len(d) - tail]
    try:
        clocks = [int(x) for x in open(sclk).read().split() if x.strip().isdigit()]
    except OSError:
        clocks = []
    ratio = float(np.median(last)) / (hip_ms * 1e3)
    print(f"mode {mode}  kernel {kern}")
    print(f"  rocprofv3, all {len(d)} launches   : avg {d.mean():9.1f}  median {np.median(d):9.1f}  min {d.min():9.1f} us")
    print(f"  rocprofv3, the {K} timed launches : avg {last.mean():9.1f}  median {np.median(last):9.1f}  min {last.min():9.1f} us")
    print(f"  HIP events, the same {K} launches : avg {hip_ms * 1e3:9.1f} us   (bench line: roofline.frac {j['roofline']['frac']:.3f})")
    print(f"  median(timed, rocprofv3) / HIP events = {ratio:.3f}   shader clock while busy (rocm-smi, max of {len(clocks)} samples): {max(clocks) if clocks else 'n/a'} MHz")
    ok = abs(ratio - 1.0) <= 0.05
    print("  " + ("OK: within 5 %" if ok else "FAIL: the trace does not reproduce the bench line (> 5 %)"))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
