#!/bin/bash
# librsx with the RSX_* experiment knobs compiled in (make EXPERIMENTS=1), as abtest/librsx_exp.so next to the product build:
#   RSX_LIB_PATH=abtest/librsx_exp.so RSX_RESCORE_PROF=1 python bench.py ...
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
C=$ROOT/navtech-radar-slam_amd/csrc
B=/tmp/rsx_exp_build
mkdir -p $ROOT/abtest $B
FLAGS="-DRSX_EXPERIMENTS=1 --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -I$ROOT/include -I$C -Wall -Wno-unused-function"
pids=()
for f in $C/*.hip $C/*.cpp; do
  o=$B/$(basename $f).o
  if [ ! -f $o ] || [ $f -nt $o ] || [ -n "$(find $C $ROOT/include -name '*.h' -newer $o | head -1)" ]; then
    extra=""; [ "$(basename $f)" = sc_spec.hip ] && extra="-mllvm -amdgpu-mfma-vgpr-form"
    /opt/rocm/bin/hipcc $FLAGS $extra -x hip -c $f -o $o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $B/*.o -ldl -o $ROOT/abtest/librsx_exp.so
echo built $ROOT/abtest/librsx_exp.so
