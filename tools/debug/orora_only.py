"""ORORA leg of bench.py alone (kernel experiments): python tools/debug/orora_only.py"""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import bench
print(json.dumps(bench.orora_leg(0, True)))
