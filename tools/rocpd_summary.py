#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd databases (gpurun_out/prof_<tag>/*/*.db) into a text report that is
small enough to commit under profiles/.  Usage: tools/rocpd_summary.py gpurun_out/prof_r01a [--all-grids] > profiles/...txt
(--all-grids: the counters of every grid size of a kernel, not only its largest one; per-grid durations of every kernel)"""
import glob
import os
import sqlite3
import sys


def main(root, all_grids=False):
    trace = glob.glob(os.path.join(root, "trace", "*.db"))
    if trace:
        con = sqlite3.connect(trace[0])
        print("== rocprofv3 --kernel-trace --stats (top kernels; durations in us) ==")
        print(f"{'calls':>7} {'total_us':>14} {'avg_us':>12} {'pct':>7}  name")
        for name, calls, total, avg, pct in con.execute("select name,total_calls,total_duration,average,percentage from top_kernels limit 12"):
            print(f"{calls:7d} {total:14.3f} {avg:12.3f} {pct:7.2f}  {name[:110]}")
        print("\n== per-dispatch durations of the dominant kernel (largest grids first) ==")
        q = ("select name, grid_x*grid_y*grid_z as g, count(*), avg(duration)/1000.0, min(duration)/1000.0, max(duration)/1000.0 "
             "from kernels where (name like '%_kernel%' or name like '%cen_%' or name like '%fe_%' or name like '%odo_%' or name like '%icp_%' or name like '%vg_%' or name like '%lv_%' or name like '%orora%' or name like '%pmc_%') "
             "group by name, g order by avg(duration) desc limit " + ("40" if all_grids else "8"))
        try:
            for name, grid, n, avg, mn, mx in con.execute(q):
                print(f"grid={grid:>10} launches={n:4d} avg_us={avg:12.3f} min_us={mn:12.3f} max_us={mx:12.3f}  {name[:60]}")
        except sqlite3.Error as e:
            print("  (kernels view unavailable:", e, ")")
    for db in sorted(glob.glob(os.path.join(root, "pmc*", "*.db"))):
        con = sqlite3.connect(db)
        print(f"\n== PMC pass {os.path.basename(os.path.dirname(db))} (largest-grid dispatches of each kernel only) ==")
        q = ("select kernel_name, counter_name, count(*), avg(value), grid_size, vgpr_count, lds_block_size from counters_collection c "
             + ("" if all_grids else "where grid_size = (select max(grid_size) from counters_collection d where d.kernel_name = c.kernel_name) ")
             + "group by kernel_name, " + ("grid_size, " if all_grids else "") + "counter_name order by kernel_name, grid_size, counter_name")
        for name, cname, n, avg, grid, vgpr, lds in con.execute(q):
            if not any(t in name for t in ("rsx", "cen_", "fe_", "odo_", "icp_", "vg_", "lv_", "orora", "sc_", "pmc_")):
                continue
            short = name.replace("void ", "").replace("(anonymous namespace)::", "")[:40]  # (split("::")[-1] cut kernels with a namespaced ARGUMENT type down to the argument list)
            print(f"{short:42s} {cname:24s} dispatches={n:3d} avg_per_dispatch={avg:18.1f} grid={grid} vgpr={vgpr} lds={lds}")


if __name__ == "__main__":
    main(sys.argv[1], "--all-grids" in sys.argv[2:])
