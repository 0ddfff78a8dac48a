#!/bin/bash
# A/B timing of library variants built by scripts/build_variants.sh:  scripts/ab_variants.sh tag1 tag2 ...   (on the GPU box)
# Every variant runs the bench line twice (interleaved) with the CPU baseline and the keyframe pass switched off.
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/ab
for rep in 1 2; do
  for t in "$@"; do
    DMSA_LIB_PATH=$R/dmsa_lidar_slam_amd/csrc/variants/libdmsa_hip_$t.so timeout 120 python $R/bench.py --steps 200 --warmup 5 --cpu-iters 0 --keyframe-steps 0 ${BENCH_ARGS} 2>/dev/null < /dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$t', d['value'], d['ms_per_step'], d['stage_ms_per_step'])" | tee -a $R/gpurun_out/ab/result.txt
  done
done
