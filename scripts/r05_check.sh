#!/bin/bash
# quick GPU check of a build: the GPU tests, one bench line, per-kernel statistics of a short run
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-chk}
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -q -x --timeout=600 ${PYTEST_ARGS} > $OUT/pytest.log 2>&1
tail -15 $OUT/pytest.log
timeout 300 python bench.py --steps 20 --warmup 5 --cpu-iters 0 > $OUT/bench.json 2> $OUT/bench.err < /dev/null
tail -c 1500 $OUT/bench.json
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- python $R/bench.py --steps 24 --warmup 3 --cpu-iters 0 --keyframe-steps 0 > $OUT/stats.log 2>&1 < /dev/null
python $R/scripts/summarize_profile.py stats $(find $OUT/stats -name "*results.db" | head -1) > $OUT/kernel_stats.txt 2>&1
head -40 $OUT/kernel_stats.txt
