"""the layout_emulation leg of bench.py alone on a random binary DB (per-rank compute of every Q x S / Gf layout, emulated on
one GPU).  Usage: python tools/bench_layouts.py [n_db]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from navtech_radar_slam_amd import synth  # noqa: E402

if __name__ == "__main__":
    n_db = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    db = synth.random_descriptors(1234, n_db, binary=True)
    q = synth.random_descriptors(4321, 8192, binary=True)
    r = bench.layout_emulation_leg(0, db, q, n_db - 30, 10, reps=3)
    print(json.dumps({"per_rank_ms": r["per_rank_ms_per_step"], "db_shards_only": r["compute_speedup_db_shards_only"],
                      "best": r["compute_speedup_of_best_layout"]}))
