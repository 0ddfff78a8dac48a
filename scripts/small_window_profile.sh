#!/bin/bash
# The small-window operating points (BASELINE.json configs 2 and 5) under rocprofv3: launches per iteration, sum of kernel time against
# wall time per iteration.   scripts/small_window_profile.sh r05  ->  gpurun_out/r05/small_window_{imu,rosette}.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-r05}
mkdir -p $OUT
for w in imu rosette; do
  STEPS=60
  timeout 200 python $R/bench.py --workload small_$w --steps $STEPS --no-extras > $OUT/small_$w.plain.json 2>/dev/null < /dev/null
  timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/small_$w -o s -- python $R/bench.py --workload small_$w --steps $STEPS --no-extras > $OUT/small_$w.json 2> $OUT/small_$w.err < /dev/null
  python - "$OUT" "$w" "$STEPS" "$R" > $OUT/small_window_$w.txt <<'PY'
import glob, json, sqlite3, sys
out, w, steps, root = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
db = sqlite3.connect(glob.glob(f"{out}/small_{w}/**/*results.db", recursive=True)[0])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
plain = json.loads(open(f"{out}/small_{w}.plain.json").read().strip().splitlines()[-1])["small_window"][w]
iters = steps + 5  # the run: 5 warm-up iterations + the timed ones, nothing else (--no-extras)
work = [(n, s, e) for n, s, e in rows if "k_sync_wait" not in n]
waits = [(n, s, e) for n, s, e in rows if "k_sync_wait" in n]
ksum = sum(e - s for _, s, e in work) / 1e3
# union of the busy intervals (several streams overlap)
iv = sorted((s, e) for _, s, e in work)
busy, cur_s, cur_e = 0, None, None
for s, e in iv:
    if cur_e is None or s > cur_e:
        busy += (cur_e - cur_s) if cur_e is not None else 0
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += (cur_e - cur_s) if cur_e is not None else 0
print(f"# small window '{w}': {plain['shape']}; {plain['gaussians']} Gaussians, {plain['memberships']} memberships, P = {plain['params']}")
print(f"# unprofiled: {plain['value']} iterations/s = {plain['ms_per_step'] * 1e3:.0f} us per iteration")
print(f"# profiled run of {iters} iterations:")
print(f"kernel launches per iteration (without the one-wave stream waits): {len(work) / iters:.1f}   (+ {len(waits) / iters:.1f} k_sync_wait)")
print(f"sum of kernel durations per iteration: {ksum / iters:.0f} us     GPU busy (union over the streams) per iteration: {busy / 1e3 / iters:.0f} us")
print(f"mean kernel duration: {ksum / max(1, len(work)):.1f} us")
agg = {}
for n, s, e in work:
    k = n.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("dmsa::", "")
    a = agg.setdefault(k, [0, 0])
    a[0] += 1; a[1] += e - s
print("kernel                                              per_iter   avg_us  us_per_iter")
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
    print(f"{k[:50]:50s} {c / iters:8.2f} {t / c / 1e3:8.2f} {t / 1e3 / iters:10.1f}")
PY
  cat $OUT/small_window_$w.txt | head -12
done
