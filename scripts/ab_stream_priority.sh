R=$GRAFT_REPO_ROOT; cd /tmp
for rep in 1 2; do for p in 0 1 2 4 6; do
  echo -n "window prio=$p: "; DMSA_DEBUG=stream_priority=$p timeout 120 python $R/bench.py --steps 200 --warmup 5 --cpu-iters 0 --keyframe-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"
done; done
for rep in 1 2; do for p in 0 1 2 4 6; do
  echo -n "keyframes prio=$p: "; DMSA_DEBUG=stream_priority=$p timeout 120 python $R/bench.py --workload keyframes --map-frames 0 --frames 32 --steps 20 --cpu-iters 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"
done; done
