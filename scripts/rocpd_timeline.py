#!/usr/bin/env python
"""Device-side timeline of a rocprofv3 --kernel-trace capture (rocpd database): every dispatch in start order with its duration and the
gap to the end of the latest-ending dispatch before it, then -- per window between two launches of the kernel named ANCHOR -- the sum of the
kernel durations on the critical chain, the sum of the gaps and the window length.
    python scripts/rocpd_timeline.py run.db [anchor=k_seg_y0] [first window to print=40] [windows=3]"""
import sqlite3
import sys

path = sys.argv[1]
anchor = sys.argv[2] if len(sys.argv) > 2 else "k_seg_y0"
w0 = int(sys.argv[3]) if len(sys.argv) > 3 else 40
nw = int(sys.argv[4]) if len(sys.argv) > 4 else 3
cur = sqlite3.connect(path).cursor()
rows = list(cur.execute("select name, start, end from kernels order by start"))
short = lambda n: n.split("(")[0].replace("void ", "").replace("bhip::", "")[:46]
idx = [k for k, r in enumerate(rows) if anchor in r[0]]
print(f"{len(rows)} dispatches, {len(idx)} windows of '{anchor}'")
lens, ksum, gsum = [], [], []
for w in range(len(idx) - 1):
    a, b = idx[w], idx[w + 1]
    t_end = rows[a][1]
    tot_k = tot_g = 0.0
    for k in range(a, b):
        name, st, en = rows[k]
        gap = (st - t_end) / 1e3
        if w0 <= w < w0 + nw:
            print(f"  w{w:<4} {short(name):46s} start +{(st - rows[a][1]) / 1e3:9.1f} us  dur {(en - st) / 1e3:8.1f}  gap {gap:7.1f}")
        if gap > 0:
            tot_g += gap
            tot_k += (en - st) / 1e3
        else:   # overlapped with an earlier kernel (the statistics pass on the second stream): only what sticks out counts
            tot_k += max(0.0, (en - t_end) / 1e3)
        t_end = max(t_end, en)
    lens.append((rows[b][1] - rows[a][1]) / 1e3); ksum.append(tot_k); gsum.append(tot_g)
import statistics as st
sel = slice(w0, None)
print(f"windows {w0}..: length median {st.median(lens[sel]):.1f} us, kernels on the chain {st.median(ksum[sel]):.1f}, gaps {st.median(gsum[sel]):.1f}")
