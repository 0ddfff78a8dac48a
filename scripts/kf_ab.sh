#!/bin/bash
# A/B of debug switches (DMSA_DEBUG) on the small-window and keyframe workloads:  scripts/kf_ab.sh "merge_sort=0" "merge_sort=1" ...
cd $GRAFT_REPO_ROOT
for opt in "" "$@"; do
  for w in small_imu small_rosette; do
    DMSA_DEBUG=$opt python bench.py --workload $w --steps 200 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1])['small_window']; k=[x for x in d if x!='note'][0]; print('$w[$opt]', d[k]['value'], d[k]['ms_per_step'])"
  done
  DMSA_DEBUG=$opt python bench.py --steps 200 --warmup 5 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('window[$opt]', d['value'], d['ms_per_step'])"
  for f in 8 32; do
  DMSA_DEBUG=$opt python bench.py --workload keyframes --map-frames 0 --frames $f --steps 30 --warmup 3 --cpu-iters 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('kf$f[$opt]', d['value'], d['ms_per_step'])"
  done
done
