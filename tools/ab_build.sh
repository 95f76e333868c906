#!/bin/bash
# Build a variant of the library for A/B timing on one GPU box:
#   tools/ab_build.sh <name> "<extra hipcc flags>" [file.hip ...]
#     -> sara_amd/lib/ab/lib_<name>.so
# then  SARA_HIP_SIFT_LIB=sara_amd/lib/ab/lib_<name>.so python bench.py ...
# Only the listed sources are recompiled with the extra flags (default: all);
# the other objects come from the regular build (run `make` first).
set -e
cd "$(dirname "$0")/../sara_amd/csrc"
name=$1; extra=$2; shift; shift
files="$*"
[ -z "$files" ] && files="pyramid_kernels.hip feature_kernels.hip keypoint_kernels.hip match_kernels.hip sift_context.cpp sift_detect.cpp"
flags="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -fvisibility=hidden $extra"
mkdir -p ../lib/ab /tmp/ab_$name
objs=""
for f in pyramid_kernels.hip feature_kernels.hip keypoint_kernels.hip match_kernels.hip match_mfma.hip graph_launcher.cpp sift_schedule.cpp sift_context.cpp sift_detect.cpp sift_results.cpp sift_operators.cpp sift_comm.cpp sift_match.cpp; do
  o=${f%.*}.o
  if echo " $files " | grep -q " $f "; then
    x=""; { [ "$f" = match_kernels.hip ] || [ "$f" = feature_kernels.hip ] || [ "$f" = pyramid_kernels.hip ]; } && x="-fno-slp-vectorize"
    /opt/rocm/bin/hipcc $flags $x -c -o /tmp/ab_$name/$o $f &
    objs="$objs /tmp/ab_$name/$o"
  else
    objs="$objs $o"
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/ab/lib_$name.so $objs -ldl -lpthread
echo built sara_amd/lib/ab/lib_$name.so
