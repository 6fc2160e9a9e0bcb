"""keyframe stream: one descriptor build (sc_insert_kernel) per keyframe + a detection at every 4th (profiling runs:
tools/prof_six.sh).  Usage: python tools/bench_insert.py [n_keyframes]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from navtech_radar_slam_amd import scancontext, synth  # noqa: E402

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    rng = np.random.default_rng(7)
    clouds = [synth.radar_cloud(rng, n_points=1200, binary_z=True) for _ in range(64)]
    m = scancontext.SCManager(sc_dist_thres=0.45, capacity_hint=n + 8)
    t0 = time.perf_counter()
    loops = 0
    for i in range(n):
        m.makeAndSaveScancontextAndKeys(clouds[i % 64])
        if i % 4 == 3:
            loops += m.detectLoopClosureID(mode=scancontext.MODE_EXHAUSTIVE, full=True)[0] >= 0
    dt = time.perf_counter() - t0
    print(f"{n} keyframes, {n / dt:.0f} keyframes/s (a detection at every 4th), {loops} loops")
