#!/bin/bash
# k_gauss_fit_all by size class (profiling-only switch fit_classes: results of the run are wrong unless 7)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-fitcls}
mkdir -p $OUT
for m in 7 1 2 4; do
  DMSA_DEBUG=fit_classes=$m timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/m$m -o s -- python $R/bench.py --steps 12 --warmup 3 --no-extras > $OUT/m$m.log 2>&1 < /dev/null
  echo "fit_classes=$m" >> $OUT/summary.txt
  python $R/scripts/summarize_profile.py stats $(find $OUT/m$m -name "*results.db" | head -1) 2>/dev/null | grep -E "k_gauss_fit|k_size_classes" >> $OUT/summary.txt
done
cat $OUT/summary.txt
