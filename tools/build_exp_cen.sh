#!/bin/bash
# abtest/librsx_cen.so: librsx with cen2019.hip compiled with -DRSX_EXPERIMENTS (RSX_CEN_CFG = 1: 256 x 16, 2: 512 x 8 row blocks)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
C=$ROOT/navtech-radar-slam_amd/csrc
make -C $C -j8 > /dev/null
mkdir -p $ROOT/abtest /tmp/rsx_exp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -I$ROOT/include -I$C -DRSX_EXPERIMENTS=1 -x hip -c $C/cen2019.hip -o /tmp/rsx_exp/cen2019.hip.o
OBJS=$(ls $C/build/*.o | grep -v "cen2019.hip.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/rsx_exp/cen2019.hip.o -ldl -o $ROOT/abtest/librsx_cen.so
echo built $ROOT/abtest/librsx_cen.so
