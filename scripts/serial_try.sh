#!/bin/bash
# usage: serial_try.sh "ENV=.. ENV=.." ...   one bench run of the reference-order path per argument (block-0 clocks + launch time)
for cfg in "$@"; do
  echo "== $cfg"
  env $cfg DMSA_SERIAL_DEBUG=1 timeout 120 python bench.py --steps 6 --warmup 2 --cpu-iters 0 --keyframe-steps 0 2>/tmp/err.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('it/s', d['value'], 'ms', d['ms_per_step'], 'avg_launch_ms', d['roofline']['avg_launch_ms'])"
  grep "serial dbg" /tmp/err.log | tail -2
done
