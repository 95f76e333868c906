# MFMA tile kernel timing experiments: which phase costs what
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
D=$R/gpurun_out/prof
mkdir -p $D
cd $R
for skip in 0 1 2 4 3 5 6; do
  SARA_HIP_MATCH_SKIP=$skip rocprofv3 --kernel-trace --stats --output-format csv -d $D -o skip$skip -- python tools/match_probe.py --no-child > /dev/null 2>&1
  python - <<PY
import csv
rows = list(csv.DictReader(open("$D/skip${skip}_kernel_stats.csv")))
print("skip $skip:", "  ".join("%s %.1f us" % (r["Name"].split("::")[-1][:22], float(r["AverageNs"]) / 1e3) for r in rows if "mfma_tiles" in r["Name"]))
PY
done
