#!/bin/bash
# only the PMC passes of scripts/profile_round.sh (into the same gpurun_out/<round>):  scripts/pmc_passes.sh r03
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-r01}
mkdir -p $OUT
KRE="${KRE:-k_residuals_chain|k_residuals_small}"
run() {
  DMSA_DEBUG=device_sync=0 timeout 120 rocprofv3 --kernel-trace --pmc $2 --kernel-include-regex "$KRE" --output-format csv -d $OUT/$1 -o $1 -- python $R/bench.py --steps 3 --warmup 1 --cpu-iters 0 --keyframe-steps 0 > $OUT/$1.log 2>&1 < /dev/null
}
run p1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
run p2 "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
run p3 "FETCH_SIZE"
run p4 "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"
ls -la $OUT/p2
