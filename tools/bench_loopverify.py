"""the loop-verification and ICP legs of bench.py alone (profiling runs: tools/prof_lv.sh)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

if __name__ == "__main__":
    out = {"loop_verify": bench.loop_verify_leg(0)}
    if len(sys.argv) < 2 or sys.argv[1] != "lv":
        out["icp"] = bench.icp_leg(0)
    print(json.dumps(out))
