run() { echo "== $*"; env "$@" python bench.py --steps 10 --warmup 2 --cpu-frames 0 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(round(d['value']/1e6,2), round(d['ms_per_step'],3), d.get('stage_ms_per_step'))
"; }
SARA_HIP_FUSE_GRADIENT=1 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_full_size.py -m gpu -x -q 2>&1 | tail -3
for rep in 1 2; do
run A=1
run SARA_HIP_SIDE_GRADIENT=0
run SARA_HIP_FUSE_GRADIENT=1
done
