# Kernel statistics of the matcher (both producers) -> gpurun_out/prof/<tag>_match_*.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r03}
D=$R/gpurun_out/prof
mkdir -p $D
cd $R
python tools/match_probe.py 2>&1 | grep ratio > $D/${TAG}_match_probe.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $D -o ${TAG}_match_mfma -- python tools/match_probe.py --no-child > /dev/null 2>&1
SARA_HIP_MATCH=exhaustive rocprofv3 --kernel-trace --stats --output-format csv -d $D -o ${TAG}_match_exh -- python tools/match_probe.py --no-child > /dev/null 2>&1
cat $D/${TAG}_match_probe.txt
python - <<PY
import csv
for name in ("mfma", "exh"):
    print("==", name)
    rows = list(csv.DictReader(open("$D/${TAG}_match_%s_kernel_stats.csv" % name)))
    for r in rows[:16]:
        if "sara_hip" in r["Name"] or "mfma" in r["Name"] or "kernel" in r["Name"]:
            print("%-64s calls %6s avg %10.1f us  %5s %%" % (r["Name"][:64], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
