run() { echo "== $*"; env "$@" python bench.py --steps 8 --warmup 2 --cpu-frames 0 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'], d.get('stage_ms_per_step') or d['config'].get('stage_ms_per_step') or [k for k in d])
"; }
run SARA_HIP_STREAMS=0
run A=1
