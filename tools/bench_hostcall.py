"""one exhaustive query through the host-buffer entry (rsx_sc_query), microseconds per call.  Usage: python tools/bench_hostcall.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from navtech_radar_slam_amd import scancontext, synth  # noqa: E402

if __name__ == "__main__":
    pool = synth.random_descriptors(5, 16, binary=True)
    for n in (1000, 10000, 100000):
        h = scancontext.SCManager(capacity_hint=n + 8)
        h.add_descriptors_f32(synth.random_descriptors(77, n, binary=True))
        for i in range(10):
            h.query(pool[i % 16:i % 16 + 1], k=1, n_eligible=n - 30)
        best = 1e9
        for rep in range(5):
            t0 = time.perf_counter()
            for i in range(50):
                h.query(pool[i % 16:i % 16 + 1], k=1, n_eligible=n - 30)
            best = min(best, (time.perf_counter() - t0) / 50 * 1e6)
        print(f"N = {n}: {best:.1f} us per host call")
        h.close()
