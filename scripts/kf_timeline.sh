#!/bin/bash
# host timeline of the keyframe pass for a few solve-thread caps
cd $GRAFT_REPO_ROOT
for t in 2 4 6 8 12 16; do
  echo "== DMSA_SOLVE_THREADS=$t"
  DMSA_SOLVE_THREADS=$t DMSA_HOST_TIMELINE=1 timeout 200 python bench.py --workload keyframes --frames 32 --steps 6 --warmup 1 --cpu-iters 0 2>&1 | grep "host timeline" | tail -2 | sed 's/.*sync#3 wait [0-9]* | //'
done
