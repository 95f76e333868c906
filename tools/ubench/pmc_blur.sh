# usage: pmc_blur.sh "<ONLY name>" <tag> "<counters>"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc
ONLY="$1" timeout 120 rocprofv3 --kernel-trace --pmc $3 --output-format csv -d $R/gpurun_out/pmc -o $2 -- $R/tools/ubench/blur_limits 64 1920 1080 > $R/gpurun_out/pmc/$2.log 2>&1
tail -3 $R/gpurun_out/pmc/$2.log
