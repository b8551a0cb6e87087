#!/usr/bin/env python
"""Per launch-shape statistics of one kernel in a rocprofv3 rocpd database (development aid).
    python profiles/rocpd_gemm_shapes.py results.db gemm_general_kernel"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
pat = f"%{sys.argv[2]}%"
cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
gx, gy, gz = [c for c in cols if c.startswith("grid")][:3]
q = (f"select {gx}, {gy}, {gz}, count(*), avg(end - start) / 1000.0, sum(end - start) / 1000.0 from kernels "
     f"where name like ? group by {gx}, {gy}, {gz} order by 6 desc")
print(f"{'grid':>20} {'calls':>6} {'avg_us':>9} {'total_us':>10}")
for a, b, c, n, avg, tot in con.execute(q, (pat,)):
    print(f"{str((a, b, c)):>20} {n:6d} {avg:9.2f} {tot:10.1f}")
