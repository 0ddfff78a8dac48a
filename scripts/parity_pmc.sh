#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/parity_pmc
mkdir -p $OUT
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $grp --kernel-include-regex "k_residuals_mirror_rows" --output-format csv -d $OUT/$tag -o c -- python $R/bench.py --mirror --host-tables --steps 3 --warmup 1 --cpu-iters 0 --keyframe-steps 0 > $OUT/$tag.log 2>&1
done
python - <<PY
import csv, glob, collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        acc[r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for g in sorted(acc):
    print("grid", g, {k: round(sum(v)/len(v),1) for k,v in sorted(acc[g].items())})
PY
