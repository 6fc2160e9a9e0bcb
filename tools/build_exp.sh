#!/bin/bash
# librsx with the host-side experiment knobs compiled in (sc_api.cpp with -DRSX_EXPERIMENTS: RSX_SC_HOST_PIECES,
# RSX_SC_FIRST_TARGET, RSX_SC_FILTER ...): abtest/librsx_exp.so.  Kernels are the shipped objects.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
C=$ROOT/navtech-radar-slam_amd/csrc
make -C $C -j8 > /dev/null
mkdir -p $ROOT/abtest /tmp/rsx_exp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt \
  -I$ROOT/include -I$C -DRSX_EXPERIMENTS=1 -x hip -c $C/sc_api.cpp -o /tmp/rsx_exp/sc_api.cpp.o
OBJS=$(ls $C/build/*.o | grep -v sc_api.cpp.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/rsx_exp/sc_api.cpp.o -ldl -o $ROOT/abtest/librsx_exp.so
echo built $ROOT/abtest/librsx_exp.so
