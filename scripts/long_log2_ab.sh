#!/bin/bash
# Where should the latency tier begin?  Builds serial_kernels.hip with -DDMSA_LONG_LOG2=<k> (Gaussians with >= 2^k members get the ten-wave
# workgroups of k_residuals_chain<8,true,128>) on the GPU box and prints the bench rates:  scripts/long_log2_ab.sh 11 13
cd $GRAFT_REPO_ROOT/dmsa_lidar_slam_amd/csrc
cp libdmsa_hip.so /tmp/libdmsa_hip.so.keep
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -pthread -Wall -Wno-unused-function -Wno-unused-result"
rates() {
  cd $GRAFT_REPO_ROOT
  python bench.py --steps 200 --warmup 5 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 window', d['value'], d['ms_per_step'])"
  python bench.py --workload keyframes --map-frames 0 --frames 32 --steps 30 --warmup 3 --cpu-iters 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 kf32', d['value'], d['ms_per_step'])"
  cd $GRAFT_REPO_ROOT/dmsa_lidar_slam_amd/csrc
}
rates "log2=12(shipped)"
for k in "$@"; do
  touch serial_kernels.hip
  make -j8 CXXFLAGS="$FLAGS -DDMSA_LONG_LOG2=$k" 2>&1 | grep -E "error" -A3
  rates "log2=$k"
done
cp /tmp/libdmsa_hip.so.keep libdmsa_hip.so
