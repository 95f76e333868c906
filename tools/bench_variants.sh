# A/B helper: runs bench.py under different environments in ONE gpurun call (box-to-box noise is
# about 3 %, run-to-run on one box about 1 %).  Edit the `run` lines; see also tools/ab_build.sh.
run() { echo "== $*"; env "$@" python bench.py --steps 10 --warmup 2 --cpu-frames 0 --no-extras 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(round(d['value']/1e6,2), round(d['ms_per_step'],3), d.get('stage_ms_per_step'))
"; }
for rep in 1 2; do
run SARA_HIP_STREAMS=1
run A=1
done
