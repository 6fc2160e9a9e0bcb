#!/bin/bash
# librsx with the spectral filter's timing instrumentation compiled in (RSX_SPEC_PROF / RSX_SPEC_DBG): abtest/librsx_instr.so
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
C=$ROOT/navtech-radar-slam_amd/csrc
make -C $C -j8 > /dev/null
mkdir -p $ROOT/abtest /tmp/rsx_instr
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt \
  -I$ROOT/include -I$C -mllvm -amdgpu-mfma-vgpr-form -DRSX_SPEC_INSTRUMENT=1 -DRSX_EXPERIMENTS=1 $EXTRA_DEFS -x hip -c $C/sc_spec.hip -o /tmp/rsx_instr/sc_spec.hip.o
OBJS=$(ls $C/build/*.o | grep -v sc_spec.hip.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/rsx_instr/sc_spec.hip.o -o $ROOT/abtest/librsx_instr.so
echo built $ROOT/abtest/librsx_instr.so
