set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 python -m pytest -m gpu -x -q tests/test_gpu_pmc.py 2>&1 | tail -15) > gpurun_out/pmc_mfma_tests.log 2>&1
tail -15 gpurun_out/pmc_mfma_tests.log
for lib in "" abtest/librsx_pmcbase.so abtest/librsx_pmc_w2.so "" abtest/librsx_pmcbase.so; do
  echo "== lib: ${lib:-product}"
  if [ -n "$lib" ]; then export RSX_LIB_PATH=$PWD/$lib; else unset RSX_LIB_PATH; fi
  NO_ORACLE=1 timeout 300 python tools/bench_pmc.py 2>&1 | grep -v amdgpu.ids | head -3
done 2>&1 | tee gpurun_out/pmc_mfma_ab.log
