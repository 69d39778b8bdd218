#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd (.db) capture as text: per-kernel call count / average duration /
registers (from --kernel-trace --stats) and per-kernel PMC counter means (from --pmc runs).

    python scripts/rocpd_summary.py gpurun_out/prof/run_results.db [more.db ...] > profiles/rNN_xxx.txt
"""
import sqlite3
import sys


def cols(cur, table):
    return [d[0] for d in cur.execute(f"select * from {table} limit 1").description]


def main():
    for path in sys.argv[1:]:
        db = sqlite3.connect(path)
        cur = db.cursor()
        print(f"== {path}")
        try:
            rows = list(cur.execute(
                "select name, count(*), avg(duration), min(duration), max(duration), sum(duration), "
                "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(scratch_size), max(lds_size), "
                "max(grid_x), max(workgroup_x) from kernels group by name order by sum(duration) desc"))
            tot = sum(r[5] for r in rows) or 1.0
            med = {}
            for (nm,) in cur.execute("select distinct name from kernels"):
                d = sorted(x[0] for x in db.cursor().execute("select duration from kernels where name = ?", (nm,)))
                med[nm] = d[len(d) // 2] if d else 0.0
            print(f"{'kernel':<100} {'calls':>5} {'avg_us':>10} {'median_us':>10} {'min_us':>10} {'max_us':>10} {'%':>6} {'vgpr':>5} {'agpr':>5} {'sgpr':>5} "
                  f"{'scratch':>7} {'lds':>6} {'grid':>10} {'wg':>4}")
            for r in rows:
                print(f"{r[0][:100]:<100} {r[1]:>5} {r[2] / 1e3:>10.1f} {med.get(r[0], 0.0) / 1e3:>10.1f} {r[3] / 1e3:>10.1f} {r[4] / 1e3:>10.1f} {100 * r[5] / tot:>6.1f} "
                      f"{r[6]:>5} {r[7]:>5} {r[8]:>5} {r[9]:>7} {r[10]:>6} {r[11]:>10} {r[12]:>4}")
        except sqlite3.Error as e:
            print("  (no kernel table)", e)
        try:
            c = cols(cur, "counters_collection")
            namecol = "kernel_name" if "kernel_name" in c else "name"
            rows = list(cur.execute(
                f"select {namecol}, counter_name, count(*), avg(value), min(value), max(value) from counters_collection "
                f"group by {namecol}, counter_name order by {namecol}, counter_name"))
            if rows:
                print(f"{'kernel':<100} {'counter':<28} {'n':>4} {'mean':>16} {'min':>16} {'max':>16}")
                for r in rows:
                    print(f"{str(r[0])[:100]:<100} {r[1]:<28} {r[2]:>4} {r[3]:>16.1f} {r[4]:>16.1f} {r[5]:>16.1f}")
        except sqlite3.Error as e:
            print("  (no counters)", e)
        print()


if __name__ == "__main__":
    main()
