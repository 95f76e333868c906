#!/bin/bash
# A/B helper for one gpurun call: each argument is an env assignment list
# ("A=1", "SARA_HIP_SIFT_LIB=... X=2"); prints keypoints/s, ms/step and stage
# times of `python bench.py` under it.  BENCH_ARGS adds bench flags.
cd $GRAFT_REPO_ROOT
for cfg in "$@"; do
  echo "== $cfg"
  env $cfg timeout 300 python bench.py --steps 10 --warmup 3 --cpu-frames 0 --no-extras --unique-frames 16 $BENCH_ARGS 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(round(d['value']/1e6,2), round(d['ms_per_step'],3), d.get('stage_ms_per_step'))
"
done
