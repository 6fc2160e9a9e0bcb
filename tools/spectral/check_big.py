#!/usr/bin/env python3
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from navtech_radar_slam_amd import scancontext as sc, synth, _rsx
n, nq = int(sys.argv[1]) if len(sys.argv) > 1 else 10000, int(sys.argv[2]) if len(sys.argv) > 2 else 3000
descs = synth.random_descriptors(3, n, binary=True)
rng = np.random.default_rng(1)
q = np.stack([synth.rotate_descriptor(descs[int(rng.integers(0, n))], int(rng.integers(0, 60))) for _ in range(nq)])
gs = sc.SCManager(filter_kind=_rsx.KIND_SPECTRAL, capacity_hint=n); gd = sc.SCManager(filter_kind=_rsx.KIND_DIRECT, capacity_hint=n)
gs.add_descriptors_f32(descs); gd.add_descriptors_f32(descs)
ls = gs.filter_bounds(q); ld = gd.filter_bounds(q)
d = ls - ld
print("diff spectral - direct: min %.5f max %.5f mean %.5f" % (d.min(), d.max(), d.mean()))
bad = np.argwhere((d < -1.7e-3) | (d > -0.9e-3))
print("outliers:", len(bad), bad[:10].tolist())
if len(bad):
    qs = np.unique(bad[:, 0]); es = np.unique(bad[:, 1])
    print("queries", qs[:20], len(qs), "entries", es[:20], len(es))
