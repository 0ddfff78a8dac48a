#!/bin/bash
# Round profile set: bench line, rocprofv3 kernel stats, PMC passes (separate runs, kernel-trace only) for the correspondence kernels.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-r01}
mkdir -p $OUT
python $R/bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- python $R/bench.py --steps 10 --warmup 2 --cpu-iters 0 --keyframe-steps 0 > $OUT/stats.log 2>&1
KRE="${KRE:-k_residuals_chain|k_residuals_small}"
run() {
  rocprofv3 --kernel-trace --pmc $2 --kernel-include-regex "$KRE" --output-format csv -d $OUT/$1 -o $1 -- python $R/bench.py --steps 3 --warmup 1 --cpu-iters 0 --keyframe-steps 0 > $OUT/$1.log 2>&1
}
run p1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
run p2 "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
run p3 "FETCH_SIZE"
run p4 "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"
# keyframe set (BASELINE config 3/4 shape) kernel stats
rocprofv3 --kernel-trace --stats -d $OUT/kfstats -o stats -- python $R/bench.py --workload keyframes --frames 32 --steps 4 --warmup 1 --cpu-iters 0 > $OUT/kfstats.log 2>&1
# repeated bench lines
for i in 1 2 3; do python $R/bench.py --steps 20 --warmup 3 --cpu-iters 0 --keyframe-steps 0 2>/dev/null | tail -1 >> $OUT/bench_runs.jsonl; done
cat $OUT/bench.json
