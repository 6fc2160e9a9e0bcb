#!/bin/bash
# A/B: first re-scoring round size x preview on/off (librsx.so = preview, tools/spectral/v/librsx_nopreview.so = off)
for lib in navtech-radar-slam_amd/librsx.so tools/spectral/v/librsx_nopreview.so; do
  for ft in 64 32 16; do
    echo -n "$(basename $lib) ft=$ft "
    RSX_LIB_PATH=$PWD/$lib RSX_SC_FIRST_TARGET=$ft timeout 100 python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), round(d['roofline']['avg_launch_ms'],3), d['planted_loops_recovered'])"
  done
done
