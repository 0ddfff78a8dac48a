#!/bin/bash
# A/B of k_gauss_fit_all builds (scripts/build_variants.sh with SRC=dmsa_kernels): per-class kernel time (fit_classes) and bench lines
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-abfit}
shift
mkdir -p $OUT
for tag in "$@"; do
  LIB=$R/dmsa_lidar_slam_amd/csrc/variants/libdmsa_hip_$tag.so
  for m in 7 1 2 4; do
    DMSA_LIB_PATH=$LIB DMSA_DEBUG=fit_classes=$m timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/${tag}_m$m -o s -- python $R/bench.py --steps 12 --warmup 3 --cpu-iters 0 --keyframe-steps 0 > $OUT/${tag}_m$m.log 2>&1 < /dev/null
    echo "$tag fit_classes=$m $(python $R/scripts/summarize_profile.py stats $(find $OUT/${tag}_m$m -name '*results.db' | head -1) 2>/dev/null | grep -E 'k_gauss_fit_all')" >> $OUT/summary.txt
  done
  for i in 1 2; do DMSA_LIB_PATH=$LIB timeout 100 python $R/bench.py --steps 100 --warmup 5 --cpu-iters 0 --keyframe-steps 0 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', d['value'], d['ms_per_step'])" >> $OUT/summary.txt; done
done
cat $OUT/summary.txt
