#!/bin/bash
# Hardware counters of ONE kernel (regex $1) over a keyframe-neighbourhood run (or BENCH_ARGS): two passes, kernel-trace only, event
# dependencies (counter collection serialises kernels across queues, see scripts/profile_round.sh).  usage (GPU box): scripts/pmc_kernel.sh k_split_pairs
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
KRE=${1:-k_split_pairs}
ARGS=${BENCH_ARGS:---workload keyframes --map-frames 0 --frames 32 --steps 3 --warmup 1 --cpu-iters 0}
OUT=$R/gpurun_out/pmc_$KRE; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
pass() {
  DMSA_DEBUG=device_sync=0 timeout 150 rocprofv3 --kernel-trace --pmc $2 --kernel-include-regex "$KRE" --output-format csv -d $OUT/$1 -o $1 -- python $R/bench.py $ARGS > $OUT/$1.log 2>&1 < /dev/null
}
pass p1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"
pass p2 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS GRBM_GUI_ACTIVE"
python - $OUT <<'PY'
import csv, glob, sys, collections
d = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/p*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        d[(r["Kernel_Name"].split("(")[0][-30:], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in sorted(d.items()):
    print(k)
    for n, v in sorted(c.items()):
        print("   %-22s mean %.4g  max %.4g  (n=%d)" % (n, sum(v) / len(v), max(v), len(v)))
PY
