#!/bin/bash
# Tile size of the onesweep sort (512 threads x 2 / 4 / 8 / 16 pairs) on the bench window (1.51 M pairs per level), the keyframe neighbourhood
# and the two small windows: it/s, the voxelisation stage and the duration of a sort pass (rocprofv3).
#   scripts/sort_tiles_ab.sh r06  ->  gpurun_out/r06/sort_tiles.txt     
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-r06}
mkdir -p $OUT
{
echo "# onesweep tile = 512 threads x items pairs; 0 = the rule (2 up to 2^16 pairs, 4 up to 2^18, else 16)"
for items in 0 2 4 8 16; do
  opt="sort_items=$items"
  DMSA_DEBUG=$opt python $R/bench.py --steps 200 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('window[$opt]', d['value'], 'it/s')"
  DMSA_DEBUG=$opt python $R/bench.py --workload keyframes --steps 10 --cpu-iters 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('keyframe pass[$opt]', d['value'], 'it/s')"
  for w in small_imu small_rosette; do
    DMSA_DEBUG=$opt python $R/bench.py --workload $w --steps 200 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1])['small_window']; k=[x for x in d if x!='note'][0]; print('$w[$opt]', d[k]['value'], 'it/s')"
  done
  DMSA_DEBUG=$opt timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/sort_tiles_$items -o s -- python $R/bench.py --steps 40 --no-extras > /dev/null 2>&1
  python - "$OUT/sort_tiles_$items" "$items" <<'PY'
import glob, sqlite3, sys
db = sqlite3.connect(glob.glob(sys.argv[1] + "/**/*results.db", recursive=True)[0])
rows = db.execute("select name, end - start from kernels").fetchall()
agg = {}
for n, d in rows:
    if "k_sort_pass" in n or "k_sort_hist" in n or "k_leaf_segments" in n:
        k = n.replace("void ", "").replace("dmsa::", "").replace("(anonymous namespace)::", "").split("(")[0]
        a = agg.setdefault(k, [0, 0]); a[0] += 1; a[1] += d
for k, (c, t) in sorted(agg.items()):
    print(f"    window[sort_items={sys.argv[2]}] {k}: {c} launches, {t / c / 1e3:.2f} us each")
PY
done
} | tee $OUT/sort_tiles.txt
