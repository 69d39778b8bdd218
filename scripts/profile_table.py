#!/usr/bin/env python3
"""The round's profile table (profiles/README.md) from the committed summaries:  python scripts/profile_table.py r6
For every bench mode: dominant kernel, rocprofv3 avg / median / min of the timed launches and the HIP events of the same launches
(profiles/<tag>_<mode>_check.txt), the roofline fraction of that run, VALU per path-step / busy (bench.profiled_valu) and the PMC HBM
bytes per path-step (bench.profiled_traffic: FETCH_SIZE x 2 + WRITE_SIZE)."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

tag = sys.argv[1] if len(sys.argv) > 1 else bench.PROFILE_TAG
bench.PROFILE_TAG = tag
print("| mode | dominant kernel | paths | algorithmic B / path-step | rocprofv3, the timed launches: avg / **median** / min µs | HIP events, the same launches, µs | median ÷ HIP events | shader clock | roofline fraction (that run) | VALU per path-step / busy | PMC HBM bytes / path-step |")
print("|---|---|---|---|---|---|---|---|---|---|---|")
for mode, spec in bench.MODES.items():
    fn = os.path.join(ROOT, "profiles", f"{tag}_{mode}_check.txt")
    if not os.path.exists(fn):
        continue
    txt = open(fn).read()
    P = spec[4]
    kern = spec[8](P)
    m_t = re.search(r"timed launches\s*:\s*avg\s+([0-9.]+)\s+median\s+([0-9.]+)\s+min\s+([0-9.]+)", txt)
    m_h = re.search(r"HIP events.*?avg\s+([0-9.]+) us\s+\(bench line: roofline.frac ([0-9.]+)\)", txt)
    m_r = re.search(r"HIP events = ([0-9.]+)\s+shader clock.*?: (\d+) MHz", txt)
    d, mp, chains = spec[1], spec[2], spec[5]
    byt = 8 * d + 16 * mp if chains else 8 * d
    tr, _ = bench.profiled_traffic(mode, kern)
    v = bench.profiled_valu(mode, kern, P)
    cell = lambda m, k: m.group(k) if m else "—"
    print(f"| `{mode}` | `{kern.replace('bhip::', '')}` | {P:,} | {byt} | {cell(m_t, 1)} / **{cell(m_t, 2)}** / {cell(m_t, 3)} | {cell(m_h, 1)} | {cell(m_r, 1)} | {cell(m_r, 2)} MHz | "
          f"{cell(m_h, 2)} | {('%.1f / %.2f' % (v['insts_per_path_step'], v['busy_frac'])) if v else '—'} | {('%.2f' % (tr / (P * 1000.0))) if tr else '—'} |")
