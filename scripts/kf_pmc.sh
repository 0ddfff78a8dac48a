#!/bin/bash
# HBM traffic of the correspondence kernels on ONE keyframe neighbourhood (32 frames, P = 186): two PMC passes (FETCH_SIZE, WRITE_SIZE;
# --kernel-trace only, device-side stream waits off as in profile_round.sh) -> gpurun_out/<tag>/kp3, kp4;  scripts/kf_pmc.sh r05
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-r01}
mkdir -p $OUT
KRE="k_residuals_chain|k_residuals_small"
run() {
  DMSA_DEBUG=device_sync=0 timeout 150 rocprofv3 --kernel-trace --pmc $2 --kernel-include-regex "$KRE" --output-format csv -d $OUT/$1 -o $1 -- python $R/bench.py --workload keyframes --map-frames 0 --frames 32 --steps 3 --warmup 1 --cpu-iters 0 > $OUT/$1.log 2>&1 < /dev/null
}
run kp3 "FETCH_SIZE"
run kp4 "WRITE_SIZE"
ls $OUT/kp3 $OUT/kp4 | head
