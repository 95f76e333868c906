# BASELINE.json config 5: 3840x2160, 5 octaves - sweep of the blur work decomposition
# (strip segment height = "tile size" of the marching kernels; 64x32 LDS tiles of the tiled kernel).
run() { echo "== $*"; env "$@" python bench.py --width 3840 --height 2160 --octaves 5 --frames-per-gpu 16 --steps 8 --warmup 2 --cpu-frames 0 --no-extras --stage 1 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('pyramid %.3f ms  %.0f GB/s  frac %.3f' % (d['stage_ms_per_step']['pyramid'], r['achieved'], r['frac']))
"; }
run SARA_HIP_BLUR=tile
for w in 1024 2048 3072 4096 6144; do run SARA_HIP_MARCH_WAVES=$w SARA_HIP_MARCH2_WAVES=$w; done
run SARA_HIP_STREAMS=1
run A=1
