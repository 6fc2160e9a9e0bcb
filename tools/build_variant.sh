#!/bin/bash
# abtest/librsx_<name>.so = the product objects with sc_spec.hip replaced by <file> (kernel A/B on one GPU box in one gpurun call)
#   usage: tools/build_variant.sh <name> <sc_spec.hip variant> [extra -D flags]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
C=$ROOT/navtech-radar-slam_amd/csrc
NAME=$1; SRC=$2; shift; shift
make -C $C -j8 > /dev/null
mkdir -p $ROOT/abtest /tmp/rsx_var_$NAME
cp $SRC /tmp/rsx_var_$NAME/sc_spec.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt \
  -I$ROOT/include -I$C -mllvm -amdgpu-mfma-vgpr-form "$@" -x hip -c /tmp/rsx_var_$NAME/sc_spec.hip -o /tmp/rsx_var_$NAME/sc_spec.hip.o
OBJS=$(ls $C/build/*.o | grep -v sc_spec.hip.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/rsx_var_$NAME/sc_spec.hip.o -ldl -o $ROOT/abtest/librsx_$NAME.so
echo built abtest/librsx_$NAME.so
