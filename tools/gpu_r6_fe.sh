set -u
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_frontend.py -x -q 2>&1 | tail -5
for rep in 1 2 3; do
echo base; timeout 300 python tools/bench_fe_cart.py 64 0 2>&1 | grep "^flags"
for v in ${VARIANTS:-}; do echo $v; for n in 64; do RSX_LIB_PATH=$PWD/abtest/librsx_fe_$v.so timeout 300 python tools/bench_fe_cart.py $n 0 2>&1 | grep "^flags"; done; done
done
for n in 1 4 16; do timeout 300 python tools/bench_fe_cart.py $n 0 4 2>&1 | grep "^flags"; done
RSX_LIB_PATH=$PWD/abtest/librsx_fe_map16.so timeout 900 python -m pytest tests/test_gpu_frontend.py -x -q 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_odometry.py tests/test_gpu_host.py -x -q 2>&1 | tail -3
timeout 300 python tools/bench_odometry.py 8 256 3 2>&1 | grep -v amdgpu | tail -4
