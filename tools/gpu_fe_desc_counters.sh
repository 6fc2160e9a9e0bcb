set -u
cd $GRAFT_REPO_ROOT
OUT=$PWD/gpurun_out/prof_r06_fedesc; mkdir -p $OUT
( cd /tmp && export TMPDIR=/tmp
  CMD="python $GRAFT_REPO_ROOT/tools/bench_odometry.py 4 256 2"
  KR="--kernel-include-regex fe_describe|fe_cart_strip"
  timeout 300 rocprofv3 $KR --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS --kernel-trace -d $OUT/pmc1 -o pmc1 -- $CMD > $OUT/pmc1.log 2>&1
  timeout 300 rocprofv3 $KR --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum SQ_INSTS_SALU --kernel-trace -d $OUT/pmc2 -o pmc2 -- $CMD > $OUT/pmc2.log 2>&1 )
python tools/rocpd_summary.py $OUT 2>&1 | grep -E "fe_describe|fe_cart" | cut -c1-140
rm -rf gpurun_out/prof_r06*/*/*.db gpurun_out/prof_r06*/*/*/*.db 2>/dev/null
