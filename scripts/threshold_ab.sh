#!/bin/bash
# Where should the lane-per-evaluation tier end?  it/s of the small workloads (config-2 window, rosette window, keyframe sets of 8 .. 24 frames)
# at forced boundaries:  scripts/threshold_ab.sh 8 32 256     (the numbers behind the rule in voxelize_driver.cpp)
cd $GRAFT_REPO_ROOT
for t in "${@:-8 32 256}"; do
  for thr in $t; do
    opt="small_threshold=$thr"
    for w in small_imu small_rosette; do
      DMSA_DEBUG=$opt python bench.py --workload $w --steps 200 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1])['small_window']; k=[x for x in d if x!='note'][0]; print('$w[$opt]', d[k]['value'], d[k]['ms_per_step'], d[k].get('gaussians'), d[k].get('memberships'))"
    done
    for f in 8 12 16 24; do
      DMSA_DEBUG=$opt python bench.py --workload keyframes --map-frames 0 --frames $f --steps 30 --warmup 3 --cpu-iters 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('kf$f[$opt]', d['value'], d['ms_per_step'], d['config']['gaussians'])"
    done
  done
done
