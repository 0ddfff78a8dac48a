#!/bin/bash
# host-side timeline of one dmsa_optimize_resident call (debug switch trace_time; 2 = the arrival of every iteration's Gaussian counts):
# where the per-call overhead goes and what every iteration of a call costs (the host waits for the counts once per iteration, so its
# iteration period is the device's)
cd $GRAFT_REPO_ROOT
DMSA_DEBUG=trace_time=1 python bench.py --steps ${1:-20} --warmup 5 --no-extras 2>&1 | grep "^\[call\]" | tail -1
for w in 5 100; do
echo "iteration periods (us) of a 60-iteration call after a warm-up call of $w:"
DMSA_DEBUG=trace_time=2 python bench.py --steps 60 --warmup $w --no-extras 2>&1 | grep "^\[call\] 60" | python -c "
import sys,re
t=[int(x) for x in re.findall(r'counts (\d+)', sys.stdin.read())]
print(' '.join(str(b-a) for a,b in zip(t,t[1:])))"
done
