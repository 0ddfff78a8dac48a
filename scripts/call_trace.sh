#!/bin/bash
# host-side timeline of one dmsa_optimize_resident call (debug switch trace_time; 2 = the arrival of every iteration's Gaussian counts,
# with the number of Gaussians / memberships and the process-wide count of device buffers that had to grow):
# where the per-call overhead goes and what every iteration of a call costs (the host waits for the counts once per iteration, so its
# iteration period is the device's)
cd $GRAFT_REPO_ROOT
DMSA_DEBUG=trace_time=1 python bench.py --steps ${1:-20} --warmup 5 --no-extras 2>&1 | grep "^\[call\]" | tail -1
for w in 5 100; do
echo "iteration periods (us) of a 40-iteration call after a warm-up call of $w: period (Gaussians, memberships, buffers grown so far)"
DMSA_DEBUG=trace_time=2 python bench.py --steps 40 --warmup $w --no-extras 2>&1 | grep "^\[call\] 40" | python -c "
import sys,re
m=re.findall(r'\[M (\d+) Mm (\d+) grown (\d+)\] counts (\d+)', sys.stdin.read())
print(' '.join('%d(%s,%s,%s)' % (int(b[3])-int(a[3]), b[0], b[1], b[2]) for a,b in zip(m,m[1:])))"
done
