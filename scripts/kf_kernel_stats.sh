#!/bin/bash
# One 32-frame keyframe neighbourhood (BASELINE config 4 at shard size): bench value unprofiled, then per-kernel durations under rocprofv3.
# usage (GPU box, repo root): scripts/kf_kernel_stats.sh [out dir]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=${1:-$R/gpurun_out/kf}
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for i in 1 2; do timeout 120 python $R/bench.py --workload keyframes --map-frames 0 --frames 32 --steps 20 --warmup 3 --cpu-iters 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('one neighbourhood:', d['value'], 'it/s', d['ms_per_step'], 'ms')"; done
rm -rf $OUT/prof
timeout 200 rocprofv3 --kernel-trace -d $OUT/prof -o kf -- python $R/bench.py --workload keyframes --map-frames 0 --frames 32 --steps 6 --warmup 2 --cpu-iters 0 > $OUT/prof.log 2>&1 < /dev/null
python $R/scripts/rocpd_stats.py $(find $OUT/prof -name "*results.db" | head -1) > $OUT/kernel_stats.txt
head -32 $OUT/kernel_stats.txt | cut -c1-130
