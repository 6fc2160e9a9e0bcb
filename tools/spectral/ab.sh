#!/bin/bash
# A/B timing of librsx variants built into tools/spectral/v/ (RSX_LIB_PATH override); prints filter ms per variant
for so in tools/spectral/v/librsx_*.so; do
  n=$(basename $so .so)
  echo -n "$n "
  RSX_LIB_PATH=$PWD/$so python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), round(d['roofline']['avg_launch_ms'],3), d['planted_loops_recovered'])"
done
