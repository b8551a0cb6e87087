"""Development tool: the kernels of ONE blocking model() call in launch order, with the idle time before each
(rocprofv3 --kernel-trace rocpd database of `python bench.py --sync-steps --no-tiers ...`):

    python tools/kernel_gaps.py <results.db> [anchor kernel substring = logmel]

The pass shown is the last complete one between two launches of the anchor kernel."""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
anchor = sys.argv[2] if len(sys.argv) > 2 else "logmel"
views = [r[0] for r in con.execute("select name from sqlite_master where type in ('view','table')")]
if "kernels" in views:
    cols = [c[1] for c in con.execute("pragma table_info(kernels)")]
    q = "select name, start, end from kernels order by start"
else:
    raise SystemExit(f"no `kernels` view in {views}")
rows = list(con.execute(q))
idx = [i for i, r in enumerate(rows) if anchor in r[0]]
a, b = [(x, y) for x, y in zip(idx, idx[1:]) if y - x > 100][-1]   # a full pass (the decode chain alone is ~200 launches)
t0 = rows[a][1]
prev_end = None
busy = 0
print(f"{'start_us':>9} {'dur_us':>8} {'gap_us':>7}  kernel")
for name, st, en in rows[a:b]:
    gap = (st - prev_end) / 1e3 if prev_end is not None else 0.0
    busy += en - st
    print(f"{(st - t0) / 1e3:9.1f} {(en - st) / 1e3:8.1f} {gap:7.1f}  {name[:100]}")
    prev_end = max(prev_end or en, en)
span = rows[b][1] - t0
print(f"pass: {span / 1e3:.1f} us from anchor to anchor, kernels busy {busy / 1e3:.1f} us, idle {(span - busy) / 1e3:.1f} us, {b - a} launches")
