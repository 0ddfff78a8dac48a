cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/winprof
mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- python $R/bench.py --steps 10 --warmup 2 --cpu-iters 0 > $OUT/stats.log 2>&1
python $R/scripts/summarize_profile.py stats $OUT/stats/stats_results.db | head -${1:-50}
