run() { echo "== $*"; env "$@" python bench.py --steps 10 --warmup 2 --cpu-frames 0 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(round(d['value']/1e6,2), round(d['ms_per_step'],3), d.get('stage_ms_per_step'))
"; }
for rep in 1 2; do
for v in base g8 g4 e6 e8; do
run SARA_HIP_SIFT_LIB=sara_amd/lib/ab/lib_$v.so
done
run SARA_HIP_SIFT_LIB=sara_amd/lib/ab/lib_g8.so SARA_HIP_GRAD_WAVES=16384
run SARA_HIP_SIFT_LIB=sara_amd/lib/ab/lib_e6.so SARA_HIP_EXTREMA_WAVES=6144
run SARA_HIP_SIFT_LIB=sara_amd/lib/ab/lib_e8.so SARA_HIP_EXTREMA_WAVES=8192
done
