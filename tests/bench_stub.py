"""CPU stand-in for one rank's GPU shard, used ONLY by tests/test_bench_launcher.py to drive bench.py's
launcher / timing / reduction logic at world 2 on gloo (RSX_BENCH_LOCAL_BACKEND=tests.bench_stub:make).
The local search is the oracle -- allowed here, this file lives under tests/."""
from tests.test_distributed_cpu import OracleShard


def make(rank, world):
    from oracle import pyoracle as po
    return OracleShard(po, rank, world)
