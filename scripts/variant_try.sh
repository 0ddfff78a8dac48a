#!/bin/bash
# usage: variant_try.sh tag1 tag2 ...   (bench the reference-order path with each variant library, two runs each)
for tag in "$@"; do
  for rep in 1 2; do
    DMSA_LIB_PATH=$PWD/dmsa_lidar_slam_amd/csrc/variants/libdmsa_hip_$tag.so timeout 120 python bench.py --steps 10 --warmup 2 --cpu-iters 0 --keyframe-steps 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$tag', 'it/s', d['value'], 'ms', d['ms_per_step'], 'avg_launch_ms', d['roofline']['avg_launch_ms'])"
  done
done
