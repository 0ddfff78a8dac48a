cd $GRAFT_REPO_ROOT
for opt in "small_threshold=1" "small_threshold=8"; do
  for w in small_imu small_rosette; do
    DMSA_DEBUG=$opt python bench.py --workload $w --steps 200 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1])['small_window']; k=[x for x in d if x!='note'][0]; print('$w[$opt]', d[k]['value'], d[k]['ms_per_step'], d[k].get('gaussians'), d[k].get('memberships'))"
  done
  DMSA_DEBUG=$opt python bench.py --workload keyframes --map-frames 0 --frames 8 --steps 30 --warmup 3 --cpu-iters 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('kf8[$opt]', d['value'], d['ms_per_step'], d['config']['gaussians'])"
done
