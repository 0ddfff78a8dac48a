#!/bin/bash
# usage: pmc_kernel.sh <kernel-regex> <outdir-name>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$2
mkdir -p $OUT
run() {
  rocprofv3 --kernel-trace --pmc $2 --kernel-include-regex "$KRE" --output-format csv -d $OUT/$1 -o $1 -- python $R/bench.py --steps 3 --warmup 1 --cpu-iters 0 > $OUT/$1.log 2>&1
}
KRE="$1"
run p1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
run p2 "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
run p3 "FETCH_SIZE"
run p4 "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"
run p5 "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT"
