set -u
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_frontend.py -x -q 2>&1 | tail -5
for rep in 1 2; do
echo base; timeout 300 python tools/bench_fe_cart.py 64 0 4 2>&1 | grep "^flags"
for v in ${VARIANTS:-}; do echo $v; RSX_LIB_PATH=$PWD/abtest/librsx_fe_$v.so timeout 300 python tools/bench_fe_cart.py 64 0 2>&1 | grep "^flags"; done
done
timeout 300 python tools/bench_fe_cart.py 1 0 4 2>&1 | grep -v amdgpu | tail -3
