#!/bin/bash
# How often does the GPU test suite die, and where?  (The ROCm 7 runtime
# crashed in hip::Graph::UpdateStreams under hipGraphLaunch, DESIGN.md
# section 5.)  usage: [ENV=..] tools/crash_loop.sh [runs] [pytest args]
export PYTHONPATH=$PWD
runs=${1:-4}; shift
args=${@:-tests}
crashes=0
for i in $(seq $runs); do
  timeout 400 python -X faulthandler -m pytest $args -m gpu -v -p no:cacheprovider > /tmp/cl.log 2>&1
  if grep -q "Segmentation\|core dumped\|Aborted" /tmp/cl.log; then
    crashes=$((crashes+1))
    grep -B1 "Fatal Python error" /tmp/cl.log | head -3 | cut -c1-160
    grep -A3 "^Current thread\|most recent call first" /tmp/cl.log | grep "File" | head -6 | cut -c1-140
  fi
  grep -E " passed| failed" /tmp/cl.log | tail -1 | cut -c1-120
done
echo "$crashes crashes in $runs runs"
