#!/bin/bash
# PMC counters of the correspondence kernels for library variants:  scripts/ab_pmc.sh "COUNTER LIST" tag1 tag2 ...   (on the GPU box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
C="$1"; shift
KRE="${KRE:-k_residuals_chain|k_residuals_small}"
mkdir -p $R/gpurun_out/ab
for t in "$@"; do
  rm -rf /tmp/abpmc_$t
  DMSA_DEBUG=device_sync=0 DMSA_LIB_PATH=$R/dmsa_lidar_slam_amd/csrc/variants/libdmsa_hip_$t.so timeout 150 rocprofv3 --kernel-trace --pmc $C --kernel-include-regex "$KRE" --output-format csv -d /tmp/abpmc_$t/p1 -o p1 -- \
    python $R/bench.py --steps 3 --warmup 1 --cpu-iters 0 --keyframe-steps 0 > /tmp/abpmc_$t.log 2>&1 < /dev/null
  echo "== $t" | tee -a $R/gpurun_out/ab/pmc.txt
  python $R/scripts/summarize_profile.py pmc /tmp/abpmc_$t 2>&1 < /dev/null | grep -v "^#" | tee -a $R/gpurun_out/ab/pmc.txt
done
