#!/bin/bash
# usage: serial_sweep.sh "ENV=.. ENV=.." ...   one bench run of the default path per argument
for cfg in "$@"; do
  env $cfg timeout 120 python bench.py --steps 20 --warmup 3 --cpu-iters 0 --keyframe-steps 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg', 'it/s', d['value'], 'ms', d['ms_per_step'], 'avg_launch_ms', d['roofline']['avg_launch_ms'])"
done
