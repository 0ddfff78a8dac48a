#!/bin/bash
for tag in "$@"; do
  DMSA_LIB_PATH=$PWD/dmsa_lidar_slam_amd/csrc/variants/libdmsa_hip_$tag.so timeout 120 python bench.py --steps 20 --warmup 3 --cpu-iters 0 --keyframe-steps 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$tag', 'it/s', d['value'], 'ms', d['ms_per_step'], 'voxelize', d['stage_ms_per_step']['voxelize'])"
done
