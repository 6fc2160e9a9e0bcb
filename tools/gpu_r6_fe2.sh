set -u
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
echo base; timeout 300 python tools/bench_fe_cart.py 64 0 2>&1 | grep "^flags"
for v in ${VARIANTS}; do echo $v; RSX_LIB_PATH=$PWD/abtest/librsx_fe_$v.so timeout 300 python tools/bench_fe_cart.py 64 0 2>&1 | grep "^flags"; done
done
