#!/bin/bash
# abtest/librsx_<name>.so = the product objects with ONE object rebuilt from <source file> with extra flags
#   usage: tools/build_variant_obj.sh <name> <csrc file, e.g. sc_window.hip> [extra flags, e.g. -DWIN_OCC=3]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
C=$ROOT/navtech-radar-slam_amd/csrc
NAME=$1; SRC=$2; shift; shift
make -C $C -j8 > /dev/null
mkdir -p $ROOT/abtest /tmp/rsx_var_$NAME
EXTRA=""; [ "$SRC" = sc_spec.hip ] && EXTRA="-mllvm -amdgpu-mfma-vgpr-form"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt \
  -I$ROOT/include -I$C $EXTRA "$@" -x hip -c $C/$SRC -o /tmp/rsx_var_$NAME/$SRC.o
OBJS=$(ls $C/build/*.o | grep -v "/$SRC.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/rsx_var_$NAME/$SRC.o -ldl -o $ROOT/abtest/librsx_$NAME.so
echo built abtest/librsx_$NAME.so
