"""A/B of the host-buffer entry's piece plan (RSX_SC_HOST_PIECES=first[:growth_x10], experiments build abtest/librsx_exp.so),
one process per plan so that the static knob is re-read; same box, same data.  Usage: python tools/ab_host_pieces.py"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def make_data():
    """the bench's trajectory DB (10 000 keyframes) and batch (8192 scans) as descriptors -> /tmp/ab_host_data.npz"""
    import numpy as np
    sys.path.insert(0, ROOT)
    from importlib import import_module
    synth = import_module("navtech-radar-slam_amd.synth")
    sc = import_module("navtech-radar-slam_amd.scancontext")
    db_pts, db_off, q_pts, q_off, _ = synth.trajectory_keyframes(1234, 10000, 4321, 8192, binary_z=True)
    out = []
    for pts, off, n in ((db_pts, db_off, 10000), (q_pts, q_off, 8192)):
        g = sc.SCManager(capacity_hint=n + 8)
        for i in range(n):
            g.makeAndSaveScancontextAndKeys(pts[off[i]:off[i + 1]])
        out.append(g.export_descriptors_f32(0, n))
        g.close()
    np.savez("/tmp/ab_host_data.npz", db=out[0], q=out[1])


def main():
    make_data()
    plans = os.environ.get("AB_PLANS", "0 512:25 256:25 1024:25 512:20 512:30 768:22 1024:20").split()
    code = open(os.path.join(ROOT, "tools", "_ab_host_child.py")).read()
    for p in plans:
        lanes, tail = "2", "one"
        if p.endswith("@p"):  # the round-5 form: every piece its own filter / select / window / re-scoring chain
            p, tail = p[:-2], "pieces"
        if p.endswith("/1"):
            p, lanes = p[:-2], "1"
        env = dict(os.environ, RSX_SC_HOST_PIECES=p, RSX_SC_HOST_LANES=lanes, RSX_SC_HOST_TAIL=tail,
                   RSX_LIB_PATH=os.path.join(ROOT, "abtest", "librsx_exp.so"))
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd=ROOT)
        print(p, "lanes", lanes, "tail", tail, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-400:])


if __name__ == "__main__":
    main()
