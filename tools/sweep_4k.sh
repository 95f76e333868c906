# BASELINE.json config 5: 16 x 3840x2160, 5 octaves, Gaussian-pyramid stage - sweep of
# the blur work decomposition on the current kernels.  The marching kernels have no
# 2-D LDS tile: their "tile" is a strip (128 / 256 columns) x a segment of rows, set by
# the waves per launch; strips run alone or as barrier-held groups of 4 / 8; the tiled
# kernel uses 64 x 32 LDS tiles.
#   bash tools/sweep_4k.sh <tag>  -> gpurun_out/prof/<tag>_4k_sweep.txt / .json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r05}
D=$R/gpurun_out/prof
mkdir -p $D
cd $R
OUT=$D/${TAG}_4k_sweep.txt
JS=$D/${TAG}_4k_sweep.jsonl
: > $OUT; : > $JS
ARGS="--width 3840 --height 2160 --octaves 5 --frames-per-gpu 16 --unique-frames 4 --steps 8 --warmup 2 --cpu-frames 0 --no-extras --stage 1"
run() {
  label="$1"; shift
  env "$@" timeout 120 python bench.py $ARGS 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        row={'decomposition': '''$label''', 'env': '''$*''', 'pyramid_ms': round(d['stage_ms_per_step']['pyramid'],3), 'achieved_GBs': round(r['achieved']), 'frac': round(r['frac'],3)}
        print('%-58s pyramid %.3f ms  %5.0f GB/s  frac %.3f' % (row['decomposition'], row['pyramid_ms'], row['achieved_GBs'], row['frac']))
        open('$JS','a').write(json.dumps(row)+'\n')
" | tee -a $OUT
}
echo "# config 5 sweep, 16 x 3840x2160 x 5 octaves, pyramid stage (48 P B bytes / HIP-event time), one MI355X" | tee -a $OUT
run "shipped (2048-wave launches, strip groups by wave count)" A=1
run "tiled kernel, 64 x 32 LDS tiles everywhere" SARA_HIP_BLUR=tile
for w in 1024 2048 3072 4096 6144 8192; do
  run "marching, $w waves per launch" SARA_HIP_MARCH_WAVES=$w SARA_HIP_MARCH2_WAVES=$w
done
for g in 1 4 8; do
  run "marching, strip groups of $g forced (R <= 6 blurs)" SARA_HIP_STRIP_GROUP=$g
done
run "shipped kernels on one stream" SARA_HIP_STREAMS=1
run "XCD-aware placement off" SARA_HIP_XCD_MAP=0
# occupancy of the shipped configuration: resident waves per SIMD, averaged over a launch
timeout 200 env SARA_HIP_STREAMS=1 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $D -o ${TAG}_4k_occ -- python bench.py $ARGS > $D/${TAG}_4k_occ.log 2>&1
python - <<PY | tee -a $OUT
import collections, csv, re
per = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for r in csv.DictReader(open("$D/${TAG}_4k_occ_counter_collection.csv")):
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("sara_hip::", "").replace("void ", "")
    per[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if r["Counter_Name"] == "SQ_WAVES":
        dur[name].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print("# occupancy (single stream): kernel, launches, avg us, waves per launch, resident waves per SIMD (4 * SQ_WAVE_CYCLES / (SQ_BUSY_CYCLES / 32 * 1024): the counter ticks once per 4 cycles - a launch whose 2032 waves are all resident reads 1.9 of 1.98)")
for name, c in sorted(per.items(), key=lambda kv: -sum(dur[kv[0]])):
    if "blur" not in name:
        continue
    n = len(dur[name])
    cyc = sum(c["SQ_BUSY_CYCLES"]) / 32.0
    print("%-52s n=%3d  %7.1f us  %6.0f waves  %.2f waves/SIMD" % (
        name[:52], n, sum(dur[name]) / n / 1e3, sum(c["SQ_WAVES"]) / n,
        4 * sum(c["SQ_WAVE_CYCLES"]) / (cyc * 1024) if cyc else 0))
PY
python - <<PY
import json
rows=[json.loads(l) for l in open("$JS")]
json.dump({"workload": "16 x 3840x2160, 5 octaves, Gaussian-pyramid stage", "rows": rows}, open("$D/${TAG}_4k_sweep.json","w"), indent=1)
PY
