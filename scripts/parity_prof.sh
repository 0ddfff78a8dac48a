#!/bin/bash
# rocprofv3 kernel stats of the parity path (serial-order sums, host pose tables) on the bench window
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/parity
mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- python $R/bench.py --steps 10 --warmup 2 --cpu-iters 0 --keyframe-steps 0 > $OUT/stats.log 2>&1
DB=$(ls $OUT/stats/*.db $OUT/stats/*/*.db 2>/dev/null | head -1)
python $R/scripts/summarize_profile.py stats $DB > $OUT/kernel_stats.txt 2>$OUT/sum.err
tail -1 $OUT/stats.log | cut -c1-200
head -14 $OUT/kernel_stats.txt
