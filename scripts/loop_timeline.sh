#!/bin/bash
# kernel timeline of one bench iteration (window) and one keyframe-pass iteration + the host-side marks
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-tl}
mkdir -p $OUT
DMSA_DEBUG=host_timeline=1 python $R/bench.py --steps 12 --warmup 3 --cpu-iters 0 --keyframe-steps 0 > $OUT/host_tl.json 2> $OUT/host_tl.err
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- python $R/bench.py --steps 24 --warmup 3 --cpu-iters 0 --keyframe-steps 0 > $OUT/stats.log 2>&1
python $R/scripts/iteration_timeline.py $OUT/stats/*/stats_results.db 16 > $OUT/timeline.txt 2>$OUT/timeline.err || python $R/scripts/iteration_timeline.py $(find $OUT/stats -name "*results.db" | head -1) 16 > $OUT/timeline.txt 2>>$OUT/timeline.err
rocprofv3 --kernel-trace --stats -d $OUT/kfstats -o stats -- python $R/bench.py --workload keyframes --frames 32 --steps 6 --warmup 2 --cpu-iters 0 > $OUT/kfstats.log 2>&1
python $R/scripts/iteration_timeline.py $(find $OUT/kfstats -name "*results.db" | head -1) 4 > $OUT/kf_timeline.txt 2>>$OUT/timeline.err
DMSA_DEBUG=host_timeline=1 python $R/bench.py --workload keyframes --frames 32 --steps 6 --warmup 2 --cpu-iters 0 > $OUT/kf_host_tl.json 2> $OUT/kf_host_tl.err
find $OUT -name "*kernel_stats.csv" | head
