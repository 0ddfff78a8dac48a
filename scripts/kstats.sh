#!/bin/bash
# per-kernel statistics of a short window run (rocprofv3 --kernel-trace --stats) + two plain bench lines:  scripts/kstats.sh <tag> [grep pattern]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-ks}
mkdir -p $OUT
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/stats -o s -- python $R/bench.py --steps 24 --warmup 3 --no-extras > $OUT/stats.log 2>&1 < /dev/null
python $R/scripts/summarize_profile.py stats $(find $OUT/stats -name "*results.db" | head -1) > $OUT/kernel_stats.txt 2>&1
python $R/scripts/iteration_timeline.py $(find $OUT/stats -name "*results.db" | head -1) 16 > $OUT/iteration_timeline.txt 2>/dev/null
grep -E "${2:-.}" $OUT/kernel_stats.txt | head -${3:-45}
for i in 1 2; do timeout 100 python $R/bench.py --steps 200 --warmup 5 --no-extras 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'])"; done
