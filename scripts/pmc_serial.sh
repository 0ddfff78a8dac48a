#!/bin/bash
# PMC passes over the reference-order correspondence kernels (k_residuals_chain / k_residuals_small) on the bench window.
# usage: pmc_serial.sh <outdir-name> [kernel-regex]        counters in separate passes, kernel-trace only (gpurun refuses more)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-pmc_serial}
KRE="${2:-k_residuals_chain}"
mkdir -p $OUT
run() {
  rocprofv3 --kernel-trace --pmc $2 --kernel-include-regex "$KRE" --output-format csv -d $OUT/$1 -o $1 -- python $R/bench.py --steps 3 --warmup 1 --cpu-iters 0 --keyframe-steps 0 $BENCH_ARGS > $OUT/$1.log 2>&1
}
run p1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
run p2 "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
run p3 "FETCH_SIZE"
run p4 "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"
run p5 "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM"
python $R/scripts/summarize_profile.py pmc $OUT > $OUT/pmc_summary.txt 2>$OUT/pmc_summary.err
cat $OUT/pmc_summary.txt
