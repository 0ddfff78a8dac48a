cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/mirprof
mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- python $R/bench.py --mirror --host-tables --steps 4 --warmup 1 --cpu-iters 0 > $OUT/stats.log 2>&1
python $R/scripts/summarize_profile.py stats $OUT/stats/stats_results.db | head -${1:-16}
