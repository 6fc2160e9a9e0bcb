#!/usr/bin/env python3
"""Kernel timeline of the last bench step in a rocprofv3 rocpd database (kernel durations and the
gaps between consecutive dispatches).  Usage: tools/step_timeline.py <trace_results.db>"""
import re
import sqlite3
import sys


def main(db):
    con = sqlite3.connect(db)
    rows = list(con.execute("select name, grid_x*grid_y*grid_z, start, end from kernels order by start"))
    def short(n):
        m = re.search(r"([A-Za-z_0-9]+_kernel)", n)
        return (m.group(1) if m else n)[:28]
    names = [(short(n), g, (e - s) / 1000.0, s, e) for n, g, s, e in rows]
    idx = [i for i, x in enumerate(names) if "sc_filter_kernel" in x[0]]
    if not idx:
        print("no sc_filter_kernel dispatch found")
        return
    i0 = idx[-1]
    prev_end = None
    tot = 0.0
    for x in names[max(0, i0 - 2):i0 + 12]:
        gap = (x[3] - prev_end) / 1000.0 if prev_end else 0.0
        print(f"{x[0]:30s} grid={x[1]:8d} dur_us={x[2]:9.1f} gap_before_us={gap:7.1f}")
        prev_end = x[4]
        tot += x[2]
    print(f"sum of durations shown: {tot:.1f} us")


if __name__ == "__main__":
    main(sys.argv[1])
