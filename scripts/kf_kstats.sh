#!/bin/bash
# per-kernel statistics + iteration timeline of one keyframe neighbourhood (32 frames, P = 186) and two plain bench lines:  scripts/kf_kstats.sh <tag> [grep pattern]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-kfks}
mkdir -p $OUT
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/stats -o s -- python $R/bench.py --workload keyframes --map-frames 0 --frames 32 --steps 6 --warmup 2 --cpu-iters 0 > $OUT/stats.log 2>&1 < /dev/null
python $R/scripts/summarize_profile.py stats $(find $OUT/stats -name "*results.db" | head -1) > $OUT/kernel_stats.txt 2>&1
python $R/scripts/iteration_timeline.py $(find $OUT/stats -name "*results.db" | head -1) 4 > $OUT/iteration_timeline.txt 2>/dev/null
grep -E "${2:-.}" $OUT/kernel_stats.txt | head -${3:-45}
for i in 1 2; do timeout 100 python $R/bench.py --workload keyframes --map-frames 0 --frames 32 --steps 30 --warmup 3 --cpu-iters 0 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('kf32', d['value'], d['ms_per_step'])"; done
timeout 200 python $R/bench.py --workload keyframes --steps 30 --warmup 2 --cpu-iters 0 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('keyframe pass', d['value'], d['ms_per_step'])"
