"""Are the filter bounds of two builds of librsx bit-identical?  (abtest/librsx_base.so = the build before a change, the in-tree
library = after.)  One process per library; random DB with empty columns, NaN and inf, queries with and without empty columns.
Usage: python tools/ab_bounds_identity.py"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from navtech_radar_slam_amd import scancontext as sc, synth
n, nq = 3000, 96
descs = synth.random_descriptors(5, n, binary=False)
descs[::7].reshape(-1, 60, 20)[:, 10:25] = 0
descs[12, 100] = np.inf
q = np.stack([synth.rotate_descriptor(descs[(i * 31) %% n], i %% 60) for i in range(nq)])
q[::3].reshape(-1, 60, 20)[:, :9] = 0
q[11, 5] = np.nan
g = sc.SCManager(capacity_hint=n)
g.add_descriptors_f32(descs)
np.save(sys.argv[1], g.filter_bounds(q))
pts = synth.keyframe_clouds(4, 10, binary_z=False, loop_frac=0.4, min_gap=2, n_points=900)[0]
h = sc.SCManager()
for c in pts:
    h.makeAndSaveScancontextAndKeys(c)
np.save(sys.argv[1] + ".ins.npy", h.filter_bounds(q[:8]))
''' % ROOT


def main():
    outs = []
    for name, lib in (("base", os.path.join(ROOT, "abtest", "librsx_base.so")), ("new", None)):
        env = dict(os.environ)
        if lib:
            env["RSX_LIB_PATH"] = lib
        out = f"/tmp/ab_bounds_{name}.npy"
        r = subprocess.run([sys.executable, "-c", CHILD, out], env=env, capture_output=True, text=True, cwd=ROOT)
        if r.returncode:
            print(name, "failed:", r.stderr[-600:])
            return 1
        outs.append(out)
    a, b = np.load(outs[0]), np.load(outs[1])
    same = np.array_equal(a.view(np.uint32), b.view(np.uint32))
    a2, b2 = np.load(outs[0] + ".ins.npy"), np.load(outs[1] + ".ins.npy")
    same2 = np.array_equal(a2.view(np.uint32), b2.view(np.uint32))
    print("bounds bit-identical:", same, a.shape, "| through the insert kernel:", same2, a2.shape)
    return 0 if same and same2 else 1


if __name__ == "__main__":
    sys.exit(main())
