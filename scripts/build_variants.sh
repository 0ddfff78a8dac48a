#!/bin/bash
# Build A/B variants of libdmsa_hip.so that differ in the -D flags of ONE kernel source (default serial_kernels.hip).
#   scripts/build_variants.sh "base:" "nopk:-DDMSA_NO_PK" ...   ->  dmsa_lidar_slam_amd/csrc/variants/libdmsa_hip_<tag>.so
# Run a variant with DMSA_LIB_PATH=<path>.  (variants/ is git-ignored; it travels with gpurun.)
set -e
cd "$(dirname "$0")/../dmsa_lidar_slam_amd/csrc"
SRC=${SRC:-serial_kernels}
mkdir -p variants
make -j8 >/dev/null
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -pthread -Wall -Wno-unused-function -Wno-unused-result"
for v in "$@"; do
  tag=${v%%:*}; defs=${v#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 $FLAGS $defs -c $SRC.hip -o variants/${SRC}_$tag.o 2>&1 | grep -E "error" -A3 || true
  objs=$(ls *.o | grep -v "^$SRC.o$" | tr '\n' ' ')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o variants/libdmsa_hip_$tag.so $objs variants/${SRC}_$tag.o
  echo "built variants/libdmsa_hip_$tag.so ($defs)"
done
