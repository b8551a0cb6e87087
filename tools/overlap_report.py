"""Development tool: from a rocprofv3 rocpd database of a forward_async run, how much of the decode chain's kernel time
overlaps encoder kernels, and how long the chain of one batch takes wall-clock.
    python tools/overlap_report.py <results.db>"""
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
cur = con.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
cols = [c[1] for c in cur.execute(f"pragma table_info({kd})")]
print(kd, cols)
ks = [t for t in tabs if "kernel_symbol" in t or "info_kernel" in t]
print(ks)
