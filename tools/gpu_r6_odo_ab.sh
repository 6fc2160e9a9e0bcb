set -u
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  echo base; timeout 300 python tools/bench_odometry.py 8 512 3 2>&1 | grep resident | tail -2
  for v in ${VARIANTS}; do echo $v; RSX_LIB_PATH=$PWD/abtest/librsx_$v.so timeout 300 python tools/bench_odometry.py 8 512 3 2>&1 | grep resident | tail -2; done
done
