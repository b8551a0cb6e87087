import sqlite3, glob, sys
con = sqlite3.connect(glob.glob(sys.argv[1] + "/*/*_results.db")[0])
rows = list(con.execute("select name, start, end, grid_x, grid_y, workgroup_x from kernels order by start"))
idx = [i for i, r in enumerate(rows) if "greedy_init" in r[0]][-1]
prev = None
for r in rows[idx: idx + 22]:
    gap = (r[1] - prev) / 1000.0 if prev else 0
    print("%-30s grid %5d x %2d wg %4d dur %6.2f us  gap %5.2f" % (r[0][23:53], r[3] // max(r[5], 1), r[4], r[5], (r[2] - r[1]) / 1000.0, gap))
    prev = r[2]
print("decode total ms", (rows[-1][2] - rows[idx][1]) / 1e6, "kernels", len(rows) - idx)
