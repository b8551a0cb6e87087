#!/usr/bin/env python
"""Per-kernel aggregate of the PMC values in a rocprofv3 rocpd database (one `--pmc` pass).

    python profiles/pmc_summary.py <results.db> [name-filter ...]
Prints, per kernel and counter: dispatches, sum and mean per dispatch.
"""
import sqlite3
import sys


def main(path, filters):
    con = sqlite3.connect(path)
    cur = con.cursor()
    cols = [c[1] for c in cur.execute("pragma table_info(counters_collection)")]
    kcol = "kernel_name" if "kernel_name" in cols else [c for c in cols if "kernel" in c and "name" in c][0]
    q = (f"select {kcol}, counter_name, count(*), sum(value), avg(value) from counters_collection "
         f"group by {kcol}, counter_name order by sum(value) desc")
    print(f"# {path.split('/')[-1]}: columns {cols}")
    print(f"{'dispatches':>10} {'sum':>16} {'mean/dispatch':>16}  counter  kernel")
    for name, counter, n, s, a in cur.execute(q):
        if filters and not any(f in name for f in filters):
            continue
        short = name if len(name) < 90 else name[:87] + "..."
        print(f"{n:10d} {s:16.6g} {a:16.6g}  {counter}  {short}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:])
