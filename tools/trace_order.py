#!/usr/bin/env python3
"""Print the kernels of the LAST forward in dispatch order with their GPU-timestamp durations."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end, grid_x, workgroup_x, lds_size, vgpr_count, accum_vgpr_count, sgpr_count from kernels order by start").fetchall()
# last forward = after the last k_nms but one
idx = [i for i, r in enumerate(rows) if "k_nms" in r[0]]
lo = idx[-2] + 1 if len(idx) >= 2 else 0
hi = idx[-1] + 1
t0 = rows[lo][1]
tot = 0
print("%-34s %9s %9s %8s %6s %7s %5s %5s" % ("kernel", "start_us", "dur_us", "grid", "wg", "lds", "vgpr", "agpr"))
for r in rows[lo:hi]:
    name = r[0].split("(")[0][:34]
    d = (r[2] - r[1]) / 1e3
    tot += d
    print("%-34s %9.1f %9.1f %8d %6d %7d %5d %5d" % (name, (r[1] - t0) / 1e3, d, r[3], r[4], r[5], r[6] or 0, r[7] or 0))
print("sum of kernel durations %.1f us; span %.1f us" % (tot, (rows[hi - 1][2] - t0) / 1e3))
