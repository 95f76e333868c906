run() { echo "== $*"; env "$@" python bench.py --steps 10 --warmup 2 --cpu-frames 0 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(round(d['value']/1e6,2), round(d['ms_per_step'],3), d.get('stage_ms_per_step'))
"; }
for rep in 1 2; do
run SARA_HIP_FUSE_BLUR_PAIR=0
run SARA_HIP_FUSE_BLUR_PAIR=1
run SARA_HIP_FUSE_BLUR_PAIR=1 SARA_HIP_PAIR_WAVES=3072
run SARA_HIP_FUSE_BLUR_PAIR=1 SARA_HIP_PAIR_WAVES=4096
done
run SARA_HIP_STREAMS=1 SARA_HIP_FUSE_BLUR_PAIR=0
run SARA_HIP_STREAMS=1 SARA_HIP_FUSE_BLUR_PAIR=1
