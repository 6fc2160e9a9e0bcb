set -u
cd $GRAFT_REPO_ROOT
OUT=$PWD/gpurun_out/prof_r06_fematch; mkdir -p $OUT
( cd /tmp && export TMPDIR=/tmp
  CMD="python $GRAFT_REPO_ROOT/tools/bench_fe_match.py"
  KR="--kernel-include-regex fe_match"
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
  timeout 300 rocprofv3 $KR --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $OUT/pmc1 -o pmc1 -- $CMD > $OUT/pmc1.log 2>&1
  timeout 300 rocprofv3 $KR --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM --kernel-trace -d $OUT/pmc2 -o pmc2 -- $CMD > $OUT/pmc2.log 2>&1
  timeout 300 rocprofv3 $KR --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA --kernel-trace -d $OUT/pmc3 -o pmc3 -- $CMD > $OUT/pmc3.log 2>&1 )
python tools/rocpd_summary.py $OUT > gpurun_out/r06_fe_match_mfma_rocprofv3.txt 2>&1; grep -E "fe_match" gpurun_out/r06_fe_match_mfma_rocprofv3.txt | cut -c1-140
rm -rf gpurun_out/prof_r06*/*/*.db gpurun_out/prof_r06*/*/*/*.db 2>/dev/null
