#!/usr/bin/env python
"""Turn a rocprofv3 rocpd database (ROCm 7.2 default output of `rocprofv3 --kernel-trace --stats`)
into the per-kernel stats table that is committed under profiles/.

    python profiles/rocpd_summary.py gpurun_out/prof/.../NNN_results.db > profiles/rNN_kernel_stats.txt
"""
import sqlite3
import sys


def main(path):
    con = sqlite3.connect(path)
    rows = list(con.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    print(f"# source: rocprofv3 --kernel-trace --stats ({path.split('/')[-1]}); durations in microseconds")
    print(f"{'calls':>7} {'total_us':>12} {'avg_us':>11} {'pct':>6}  kernel")
    for name, calls, total, avg, pct in rows:
        if len(name) > 110:
            name = name[:107] + "..."
        print(f"{calls:7d} {total:12.1f} {avg:11.2f} {pct:6.2f}  {name}")


if __name__ == "__main__":
    main(sys.argv[1])
