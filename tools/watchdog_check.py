"""The grid barrier's watchdog (csrc/rsx_grid_dev.h) on a deliberately oversubscribed device: an experiments build of icp.hip
(tools/build_variant_obj.sh wd icp.hip -DRSX_EXPERIMENTS=1 -DRSX_GRID_PATIENCE=30000000ull) launched with 600 workgroups of
1024 threads -- 344 of them cannot be resident, the first barrier can never complete.  Expected: an error status after the
patience (0.3 s here, 5 s in the product), twice in a row from the same handle (the counters were left clean), and a correct
alignment from a normally sized launch afterwards.  NOT part of the pytest suite: if the watchdog were broken this would hang
the device.  Usage: RSX_LIB_PATH=abtest/librsx_wd.so python tools/watchdog_check.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from navtech_radar_slam_amd import icp  # noqa: E402
from navtech_radar_slam_amd._rsx import RsxError  # noqa: E402
from test_oracle_icp import scene  # noqa: E402


def main():
    tgt = scene(1, 3000)
    src = (tgt[::3] + np.float32([0.3, 0.1, 0.0])).astype(np.float32)
    os.environ["RSX_ICP_WGS"] = "600"
    bad = icp.Icp()
    for attempt in range(2):
        t0 = time.perf_counter()
        try:
            bad.align(src, tgt)
            print("attempt", attempt, "UNEXPECTED: no error")
        except RsxError as e:
            print("attempt", attempt, f"gave up after {time.perf_counter() - t0:.2f} s:", str(e)[:90])
    del os.environ["RSX_ICP_WGS"]
    good = icp.Icp()
    r = good.align(src, tgt)
    print("normal launch afterwards: converged", r["converged"], "iterations", r["iterations"], "fitness %.2e" % r["fitness"])


if __name__ == "__main__":
    main()
