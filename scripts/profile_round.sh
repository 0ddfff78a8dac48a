#!/bin/bash
# Round profile set: bench line, rocprofv3 kernel stats + iteration timeline, PMC passes (separate runs, kernel-trace only) for the
# correspondence kernels, keyframe-pass stats + timeline, repeated bench lines.   scripts/profile_round.sh r03 ; then
# python scripts/make_round_profiles.py r03 (here) turns gpurun_out/r03 into profiles/r03_*.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-r01}
mkdir -p $OUT
timeout 300 python $R/bench.py --steps 200 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err < /dev/null
timeout 120 rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- python $R/bench.py --steps 24 --warmup 3 --no-extras > $OUT/stats.log 2>&1 < /dev/null
python $R/scripts/iteration_timeline.py $(find $OUT/stats -name "*results.db" | head -1) 16 > $OUT/iteration_timeline.txt 2> $OUT/timeline.err
DMSA_DEBUG=host_timeline=1 timeout 100 python $R/bench.py --steps 12 --warmup 3 --no-extras > /dev/null 2> $OUT/host_timeline.err < /dev/null
KRE="${KRE:-k_residuals_chain|k_residuals_small}"
# Counter collection serialises kernels ACROSS queues, in an order of its own: a one-wave wait of the device-side stream dependencies
# (csrc/dev_sync.h) can then get the chip before the kernel that signals it, and gives up after seconds (DMSA_ERR_HIP, no counters).
# The counters are those of the correspondence kernels, which do not care how the streams are ordered: events for these passes.
run() {
  DMSA_DEBUG=device_sync=0 timeout 120 rocprofv3 --kernel-trace --pmc $2 --kernel-include-regex "$KRE" --output-format csv -d $OUT/$1 -o $1 -- python $R/bench.py --steps 3 --warmup 1 --no-extras > $OUT/$1.log 2>&1 < /dev/null
}
run p1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
run p2 "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
run p3 "FETCH_SIZE"
run p4 "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"
# keyframe set (BASELINE config 4 at shard size): kernel stats + timeline
timeout 120 rocprofv3 --kernel-trace --stats -d $OUT/kfstats -o stats -- python $R/bench.py --workload keyframes --map-frames 0 --frames 32 --steps 6 --warmup 2 --cpu-iters 0 > $OUT/kfstats.log 2>&1 < /dev/null
python $R/scripts/iteration_timeline.py $(find $OUT/kfstats -name "*results.db" | head -1) 4 > $OUT/keyframes_iteration_timeline.txt 2>> $OUT/timeline.err
# the keyframe workload's own bench line (strong-scaling layout: the 249-frame map in 8 neighbourhoods on this one GPU)
timeout 300 python $R/bench.py --workload keyframes --steps 30 --warmup 2 --cpu-iters 0 > $OUT/bench_keyframes.json 2> $OUT/bench_keyframes.err < /dev/null
# HBM traffic of the correspondence kernels on a keyframe neighbourhood (two PMC passes)
bash $R/scripts/kf_pmc.sh ${1:-r01} > /dev/null 2>&1
# repeated bench lines
for i in 1 2 3; do timeout 100 python $R/bench.py --steps 200 --warmup 5 --no-extras 2>/dev/null < /dev/null | tail -1 >> $OUT/bench_runs.jsonl; done
tail -c 400 $OUT/bench.json
# the driver's own command (20 timed steps after 5 warm-up steps), four times
for i in 1 2 3 4; do timeout 100 python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-extras 2>/dev/null < /dev/null | tail -1 >> $OUT/bench_driver_command.jsonl; done
# the small-window operating points (configs 2 and 5)
bash $R/scripts/small_window_profile.sh ${1:-r01} > /dev/null 2>&1
# unprofiled gaps around the normal equations (keyframes P = 186, window P = 30)
DMSA_DEBUG=gap_stamps=1 timeout 100 python $R/bench.py --workload keyframes --map-frames 0 --frames 32 --steps 10 --warmup 2 --cpu-iters 0 2>&1 | grep gap_stamps | tail -2 > $OUT/gap_stamps.txt
DMSA_DEBUG=gap_stamps=1 timeout 100 python $R/bench.py --steps 10 --warmup 2 --no-extras 2>&1 | grep gap_stamps | tail -1 >> $OUT/gap_stamps.txt
