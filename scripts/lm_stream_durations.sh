cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_q; timeout 120 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_q -o t -- python -m pytest $R/tests/test_gpu_loop.py -k "device_lm_step and (186-1 or 72-1 or 129-1)" -x -q > /tmp/prof_q.log 2>&1; tail -1 /tmp/prof_q.log
python - <<'PY'
import csv,glob,collections
d=collections.defaultdict(list)
for r in csv.DictReader(open(glob.glob('/tmp/prof_q/**/*kernel_trace.csv',recursive=True)[0])):
    n=r['Kernel_Name']
    if 'lm_stream' in n: d[('tail' if 'tail' in n else 'stream', r.get('Grid_Size_X', r.get('Grid_Size','')))].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000.0)
for k,v in sorted(d.items()): print(k, 'n=%d'%len(v), sorted(round(x,1) for x in v))
PY
