#!/usr/bin/env python3
"""Instruction mix of the hottest loop of one kernel in a gfx950 assembly listing (hipcc -S --cuda-device-only).
usage: isa_loop_stats.py file.s <kernel symbol substring>
The hot loop = the backward branch spanning the most instructions."""
import collections
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and sys.argv[2] in l.split(":")[0] and ":" in l)
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
body = lines[start:end + 1]
labels = {}
for i, l in enumerate(body):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        labels[m.group(1)] = i
best = None
for i, l in enumerate(body):
    m = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", l) or re.match(r"\s+s_branch\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        n = sum(1 for k in body[labels[m.group(1)]:i] if re.match(r"\s+[vsdgb]\w+_", k))
        if best is None or n > best[0]:
            best = (n, labels[m.group(1)], i)
n, a, b = best
cnt = collections.Counter()
for k in body[a:b + 1]:
    m = re.match(r"\s+([a-z]\w+)", k)
    if m and not k.strip().startswith((";", ".")):
        cnt[m.group(1)] += 1
tot = sum(cnt.values())
valu = sum(v for k, v in cnt.items() if k.startswith("v_"))
print(f"loop lines {a}-{b}: {tot} instructions, {valu} VALU, {sum(v for k, v in cnt.items() if k.startswith('s_'))} SALU/scalar, "
      f"{sum(v for k, v in cnt.items() if k.startswith(('global_', 'buffer_', 'flat_', 'ds_')))} memory")
for k, v in cnt.most_common(60):
    print(f"  {k:28s} {v}")
