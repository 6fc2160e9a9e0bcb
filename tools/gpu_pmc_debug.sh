set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/pmc_debug.log
for st in plain pmc1 pmc2 pmc3 pmc4 pmc5 pmc6 solver_pmc; do
  timeout -s KILL ${STAGE_TIMEOUT:-45} python -u tools/debug/pmc_stages.py $st >> gpurun_out/pmc_debug.log 2>&1
  echo "stage $st rc=$?" >> gpurun_out/pmc_debug.log
done
cat gpurun_out/pmc_debug.log | grep -v amdgpu.ids
