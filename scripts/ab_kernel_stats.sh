#!/bin/bash
# per-kernel durations of library variants (scripts/build_variants.sh):  scripts/ab_kernel_stats.sh tag1 tag2 ...   (on the GPU box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
KRE="${KRE:-k_residuals|k_loop|k_normal}"
mkdir -p $R/gpurun_out/ab
for t in "$@"; do
  rm -rf /tmp/abks_$t
  DMSA_LIB_PATH=$R/dmsa_lidar_slam_amd/csrc/variants/libdmsa_hip_$t.so timeout 150 rocprofv3 --kernel-trace --stats -d /tmp/abks_$t -o s -- \
    python $R/bench.py --steps 24 --warmup 3 --cpu-iters 0 --keyframe-steps 0 ${BENCH_ARGS} > /tmp/abks_$t.log 2>&1 < /dev/null
  f=$(find /tmp/abks_$t -name "*results.db" | head -1)
  echo "== $t" | tee -a $R/gpurun_out/ab/kstats.txt
  if [ -n "$f" ]; then python $R/scripts/summarize_profile.py stats "$f" 2>/dev/null < /dev/null | grep -E "$KRE" | tee -a $R/gpurun_out/ab/kstats.txt; else tail -5 /tmp/abks_$t.log; fi
done
