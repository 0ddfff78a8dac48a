#!/bin/bash
# Kernel durations of the device LM solve (k_loop_lm_stream / k_loop_lm_stream_tail / k_loop_lm_panels) for library variants built by
#   SRC=loop_kernels scripts/build_variants.sh "s8x8:" "s16x4:-DDMSA_STREAM_W=16 -DDMSA_STREAM_WORKERS=4" ...
# usage (on the GPU box, from the repo root): scripts/ab_lm_stream.sh s8x8 s16x4 ...
# Runs the P = 72 / 129 / 186 cases of tests/test_gpu_loop.py::test_device_lm_step_equals_the_oracle_step under rocprofv3 --kernel-trace
# and prints min / mean per (kernel, grid size).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp; export TMPDIR=/tmp
for v in "$@"; do
  rm -rf /tmp/prof_$v
  DMSA_LIB_PATH=$R/dmsa_lidar_slam_amd/csrc/variants/libdmsa_hip_$v.so timeout 120 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$v -o t -- \
      python -m pytest $R/tests/test_gpu_loop.py -k "device_lm_step and (186-1 or 72-1 or 129-1)" -x -q > /tmp/prof_$v.log 2>&1
  f=$(find /tmp/prof_$v -name "*kernel_trace.csv" | head -1)
  python - "$f" "$v" <<'PY'
import csv, sys, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "lm_stream" in n or "lm_panels" in n:
        name = "tail" if "tail" in n else "stream" if "lm_stream" in n else "panels"
        d[(name, r.get("Grid_Size_X", r.get("Grid_Size", "")))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0)
for k, v in sorted(d.items()):
    print(sys.argv[2], k, "n=%d" % len(v), "min %.1f mean %.1f us" % (min(v), sum(v) / len(v)))
PY
done
