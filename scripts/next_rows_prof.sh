#!/bin/bash
# rocprofv3 kernel stats of the SURVEY 8(f) rows (addStaticPoints, preProcess, keyframe creation) at the sizes of scripts/static_time.py
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/next_rows
mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- python $R/scripts/static_time.py > $OUT/stats.log 2>&1
DB=$(ls $OUT/stats/*/*.db $OUT/stats/*.db 2>/dev/null | head -1)
python $R/scripts/summarize_profile.py stats $DB > $OUT/kernel_stats.txt 2>$OUT/sum.err
tail -1 $OUT/stats.log | cut -c1-300
head -40 $OUT/kernel_stats.txt
