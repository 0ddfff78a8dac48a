#!/bin/bash
# What does the throughput tier (k_residuals_chain<4,false,32>) write 12 MB per launch for?  WRITE_SIZE of the shipped build (64 VGPRs at
# 8 waves per SIMD, a few registers spilled) against a build at 6 waves per SIMD (no spills), and the bench rate of both.
cd $GRAFT_REPO_ROOT/dmsa_lidar_slam_amd/csrc
cp libdmsa_hip.so /tmp/libdmsa_hip.so.keep
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -pthread -Wall -Wno-unused-function -Wno-unused-result"
pass() {
  cd /tmp && export TMPDIR=/tmp
  rm -rf /tmp/mw_$1
  DMSA_DEBUG=device_sync=0 timeout 150 rocprofv3 --kernel-trace --pmc WRITE_SIZE --kernel-include-regex "k_residuals_chain" --output-format csv -d /tmp/mw_$1 -o w -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-extras > /tmp/mw_$1.log 2>&1 < /dev/null
  python - "$1" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for f in glob.glob(f"/tmp/mw_{sys.argv[1]}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]; n = n[n.index("k_"):].split("(")[0]
        agg[(n, "Jacobian" if int(r["Grid_Size"]) // int(r["Workgroup_Size"]) > 2600 else "other")].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print(sys.argv[1], k, "WRITE_SIZE %.2f MB per launch (%d launches)" % (sum(v) / len(v) * 1024 / 1e6, len(v)))
PY
  cd $GRAFT_REPO_ROOT && for i in 1 2; do python bench.py --steps 200 --warmup 5 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 bench', d['value'], d['ms_per_step'])"; done
  cd $GRAFT_REPO_ROOT/dmsa_lidar_slam_amd/csrc
}
pass shipped
touch serial_kernels.hip
make -j8 CXXFLAGS="$FLAGS -DDMSA_MID_WAVES=6" 2>&1 | grep -E "error" -A3
pass six_waves
cp /tmp/libdmsa_hip.so.keep libdmsa_hip.so
