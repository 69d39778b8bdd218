#!/usr/bin/env python3
"""All loops (backward branches) of one kernel in a gfx950 assembly listing, with their VALU / SALU / memory instruction counts.
usage: isa_loops.py file.s <kernel symbol substring> [min instructions]"""
import collections
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and sys.argv[2] in l.split(":")[0] and ":" in l)
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
body = lines[start:end + 1]
minn = int(sys.argv[3]) if len(sys.argv) > 3 else 20
labels = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
for i, l in enumerate(body):
    m = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", l) or re.match(r"\s+s_branch\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        a = labels[m.group(1)]
        cnt = collections.Counter()
        for k in body[a:i + 1]:
            mm = re.match(r"\s+([a-z]\w+)", k)
            if mm and not k.strip().startswith((";", ".")):
                cnt[mm.group(1)] += 1
        tot = sum(cnt.values())
        if tot < minn:
            continue
        valu = sum(v for k, v in cnt.items() if k.startswith("v_"))
        trans = sum(v for k, v in cnt.items() if re.match(r"v_(rcp|rsq|sqrt|div_|cvt_f64|cvt_i32_f64|cvt_u32_f64)", k))
        print(f"loop lines {a}-{i}: {tot} instr, {valu} VALU (of which {cnt['v_rsq_f64_e32']} rsq, {cnt['v_rcp_f64_e32']} rcp, {cnt['v_mad_u64_u32'] + cnt['v_mul_hi_u32'] + cnt['v_mul_lo_u32']} int mul), "
              f"{sum(v for k, v in cnt.items() if k.startswith('s_'))} scalar, {sum(v for k, v in cnt.items() if k.startswith(('global_', 'buffer_', 'flat_', 'scratch_')))} vmem, "
              f"{sum(v for k, v in cnt.items() if k.startswith('ds_'))} lds, barriers {cnt['s_barrier']}")
