#!/bin/bash
# Build a variant of the library for A/B timing on one GPU box:
#   tools/ab_build.sh <name> "<extra hipcc flags>"  ->  sara_amd/lib/ab/lib_<name>.so
# then  SARA_HIP_SIFT_LIB=sara_amd/lib/ab/lib_<name>.so python bench.py ...
set -e
cd "$(dirname "$0")/../sara_amd/csrc"
name=$1; shift
flags="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -fvisibility=hidden $*"
mkdir -p ../lib/ab /tmp/ab_$name
for f in pyramid_kernels.hip feature_kernels.hip match_kernels.hip; do
  /opt/rocm/bin/hipcc $flags -c -o /tmp/ab_$name/${f%.hip}.o $f &
done
/opt/rocm/bin/hipcc $flags -c -o /tmp/ab_$name/sift_context.o sift_context.cpp &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/ab/lib_$name.so /tmp/ab_$name/*.o
echo built sara_amd/lib/ab/lib_$name.so
