#!/usr/bin/env python
"""Per-DISPATCH listing of a rocprofv3 rocpd capture (--kernel-trace --pmc ...): for every dispatch of the kernels whose name
contains PATTERN, in launch order: duration and every counter (summed over its dimension instances), then -- for counters with
more than one instance -- min / max over the instances (per-channel skew).

    python scripts/rocpd_dispatches.py run.db [pattern] [block]      block > 0: also the mean of every consecutive `block` dispatches
"""
import collections
import sqlite3
import sys


def cols(cur, table):
    return [d[0] for d in cur.execute(f"select * from {table} limit 1").description]


def main():
    path = sys.argv[1]
    pat = sys.argv[2] if len(sys.argv) > 2 else "k_pc"
    block = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    cur = sqlite3.connect(path).cursor()
    kc = cols(cur, "kernels")
    print("# kernels columns:", kc)
    cc = cols(cur, "counters_collection")
    print("# counters_collection columns:", cc)
    idc = "dispatch_id" if "dispatch_id" in kc else "id"
    namec = "kernel_name" if "kernel_name" in cc else "name"
    disp = list(cur.execute(f"select {idc}, name, start, duration from kernels where name like ? order by start", (f"%{pat}%",)))
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    cidc = "dispatch_id" if "dispatch_id" in cc else idc
    for did, cname, v in cur.execute(f"select {cidc}, counter_name, value from counters_collection where {namec} like ?", (f"%{pat}%",)):
        vals[did][cname].append(v)
    names = sorted({c for d in vals.values() for c in d})
    print(f"{'#':>4} {'dispatch':>9} {'us':>10} " + " ".join(f"{n[-26:]:>26}" for n in names))
    rows = []
    for k, (did, name, start, dur) in enumerate(disp):
        r = [sum(vals[did].get(n, [0.0])) for n in names]
        rows.append((dur / 1e3, r))
        print(f"{k:>4} {did:>9} {dur / 1e3:>10.1f} " + " ".join(f"{x:>26.1f}" for x in r))
    if block > 0:
        print(f"# means over consecutive blocks of {block} dispatches (the first of a block is the warm-up: left out)")
        for b in range(0, len(rows), block):
            seg = rows[b + 1:b + block]
            if not seg:
                continue
            print(f"blk{b // block:>3} {'':>7} {sum(s[0] for s in seg) / len(seg):>10.1f} " +
                  " ".join(f"{sum(s[1][j] for s in seg) / len(seg):>26.1f}" for j in range(len(names))))
    multi = [n for n in names if any(len(d.get(n, [])) > 1 for d in vals.values())]
    if multi:
        print("# per-instance spread (min / max / max:min over the counter's dimension instances), per dispatch")
        for k, (did, name, start, dur) in enumerate(disp):
            parts = []
            for n in multi:
                v = vals[did].get(n, [])
                if v:
                    lo, hi = min(v), max(v)
                    parts.append(f"{n[-22:]}: n={len(v)} {lo:.0f}/{hi:.0f}/{(hi / lo if lo else float('inf')):.3f}")
            print(f"{k:>4} {dur / 1e3:>9.1f}  " + "  ".join(parts))


if __name__ == "__main__":
    main()
