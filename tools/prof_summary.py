#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (trace_results.db) or *_kernel_stats.csv into the
compact per-kernel table committed under profiles/.

    python tools/prof_summary.py gpurun_out/prof/trace_results.db "command line that was profiled" > profiles/rNN_x.txt
"""
import csv
import sqlite3
import sys


def short(name, n=72):
    name = name.split("(")[0] if name.startswith(("k_", "ffgpu", "void k_")) else name
    return name if len(name) <= n else name[: n - 3] + "..."


def from_db(path):
    db = sqlite3.connect(path)
    rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                      "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), "
                      "max(grid_x), max(workgroup_x) from kernels group by name, grid_x, workgroup_x order by sum(duration) desc").fetchall()
    return rows


def main():
    path = sys.argv[1]
    cmd = sys.argv[2] if len(sys.argv) > 2 else ""
    rows = from_db(path)
    total = sum(r[2] for r in rows) or 1
    print("# rocprofv3 --kernel-trace --stats summary (durations in microseconds); one row per kernel AND launch shape (grid, workgroup):")
    print("# the same kernel serves different layers / benchmark shapes, and an average over them would describe none")
    if cmd:
        print("# command:", cmd)
    print("%-72s %7s %12s %10s %10s %10s %6s %5s %5s %5s %7s %9s %5s" %
          ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct", "vgpr", "agpr", "sgpr", "lds", "grid", "wg"))
    # the 48 biggest rows, and -- wherever they rank -- the kernels the roofline figures of the bench line are about (the extras of a default run push
    # config[2]'s 54 launches below the cut: profiles/r05_d had no row for k_pw_x3t)
    keep = ("k_dw3_stream", "k_pw_x3t", "k_pw_gemm32")
    for r in rows[:48] + [r for r in rows[48:] if any(k in r[0] for k in keep)]:
        print("%-72s %7d %12.1f %10.2f %10.2f %10.2f %6.2f %5d %5d %5d %7d %9d %5d" %
              (short(r[0]), r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / total,
               r[6] or 0, r[7] or 0, r[8] or 0, r[9] or 0, r[12 - 1] or 0, r[12] or 0))


if __name__ == "__main__":
    main()
