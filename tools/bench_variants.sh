run() { echo "== $*"; env "$@" python bench.py --steps 10 --warmup 2 --cpu-frames 0 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(round(d['value']/1e6,2), round(d['ms_per_step'],3), d.get('stage_ms_per_step'))
"; }
for rep in 1 2; do
run A=1
run SARA_HIP_GRAD_MARCH_MIN_PIXELS=8000000
run SARA_HIP_GRAD_MARCH_MIN_PIXELS=30000000
run SARA_HIP_EXTREMA_MARCH_MIN_PIXELS=4000000
run SARA_HIP_EXTREMA_MARCH_MIN_PIXELS=10000000
done
