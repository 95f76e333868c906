# Kernel timeline of one steady-state single-frame call -> stdout
#   bash tools/b1_trace.sh <tag> [ENV=VALUE ...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-b1}
shift
D=$R/gpurun_out/prof
mkdir -p $D
cd $R
env "$@" rocprofv3 --kernel-trace --output-format csv -d $D -o ${TAG}_b1 -- python tools/b1_probe.py > $D/${TAG}_b1.log 2>&1
grep -v '^[WE]20' $D/${TAG}_b1.log | tail -8
python tools/b1_timeline.py $D/${TAG}_b1_kernel_trace.csv
