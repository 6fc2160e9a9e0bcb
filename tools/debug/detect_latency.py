"""Latency of the 1 Hz detector in candidate mode (tree walk) at several DB sizes: python tools/debug/detect_latency.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from navtech_radar_slam_amd import scancontext as sc, synth
for n in (1000, 10000, 100000):
    m = sc.SCManager(sc_dist_thres=0.45)
    d = synth.random_descriptors(3, n, binary=True)
    m.add_descriptors_f32(d) if hasattr(m, "add_descriptors_f32") else [m.saveScancontextAndKeys(x.astype(np.float64)) for x in d]
    t = []
    for i in range(40):
        t0 = time.perf_counter(); m.detectLoopClosureID(); t.append(time.perf_counter() - t0)
    t = np.array(t) * 1e3
    print(f"N={n}: first call (tree build) {t[0]:.2f} ms, later calls median {np.median(t[1:30]):.3f} ms, call 31 (rebuild) {t[30]:.2f} ms")
