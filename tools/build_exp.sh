#!/bin/bash
# librsx with the experiment knobs compiled in (sc_api.cpp and sc_spec.hip with -DRSX_EXPERIMENTS: RSX_SC_HOST_PIECES,
# RSX_SC_FIRST_TARGET, RSX_SC_FILTER, RSX_SPEC_XCD ...), no timing instrumentation: abtest/librsx_exp.so.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
C=$ROOT/navtech-radar-slam_amd/csrc
make -C $C -j8 > /dev/null
mkdir -p $ROOT/abtest /tmp/rsx_exp
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -I$ROOT/include -I$C -DRSX_EXPERIMENTS=1"
/opt/rocm/bin/hipcc $F -x hip -c $C/sc_api.cpp -o /tmp/rsx_exp/sc_api.cpp.o &
/opt/rocm/bin/hipcc $F -mllvm -amdgpu-mfma-vgpr-form -x hip -c $C/sc_spec.hip -o /tmp/rsx_exp/sc_spec.hip.o &
/opt/rocm/bin/hipcc $F -x hip -c $C/sc_q1.hip -o /tmp/rsx_exp/sc_q1.hip.o &
wait
OBJS=$(ls $C/build/*.o | grep -v "sc_api.cpp.o\|sc_spec.hip.o\|sc_q1.hip.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/rsx_exp/sc_api.cpp.o /tmp/rsx_exp/sc_spec.hip.o /tmp/rsx_exp/sc_q1.hip.o -ldl -o $ROOT/abtest/librsx_exp.so
echo built $ROOT/abtest/librsx_exp.so
