# How the stages of the 64 x 1080p step scale with the CUs they may use
# (SARA_HIP_CU_COUNT): decides whether a CU-partitioned overlap can pay.
cd $GRAFT_REPO_ROOT
for n in 256 224 192 160 128 96 64; do
  SARA_HIP_CU_COUNT=$n python bench.py --cpu-frames 0 --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_step']
print('CUs %3d: step %.2f ms  pyramid %.2f  extrema+gradient %.2f  orientation %.2f  descriptor %.2f' % ($n, d['ms_per_step'], s['pyramid'], s['extrema']+s['gradient'], s['orientation'], s['descriptor']))"
done
