#!/bin/bash
# First contact with a multi-GPU MI355X node (BASELINE.json
# config 4: 512 frames of 1920x1080, 64 per GPU, RCCL gather of the OERegion arrays).
# The RCCL transport of sift_comm.cpp has never run with more than one rank (no such
# node in the pool): this script is the order in which to find
# out what breaks, cheapest check first, so that the first SCALE record is a curve and
# not a debugging session.  Everything is written to gpurun_out/first_contact/.
#
#   bash tools/first_contact_8gpu.sh [N=8]
#
# Steps (each one stops the script with a message when it fails):
#   0. what the node is: devices, xGMI topology, RCCL version, HIP runtime
#   1. the library builds and loads; one-GPU smoke (parity against the CPU oracle)
#   2. -m gpu multi-rank tests on ONE device over the loopback transport (the exchange
#      logic without RCCL) - these already pass on a 1-GPU box
#   3. N = 2 with RCCL, --strict-h2h: bench.py refuses loopback when the box has enough
#      devices, verify_gather compares the BYTES that reach the root (sha256 per rank
#      shard) - both launch forms: one process per GPU (torch.distributed.run) and one
#      process driving all GPUs (sara_hip_sift_group_*, ncclCommInitAll)
#   4. N = 1, 2, 4, 8 back to back, as the driver runs them -> scale.jsonl + a table of
#      keypoints/s, speed-up over N = 1 on the SAME box, gather_verified, rccl_nranks
#   5. per-rank view of the N-GPU run: NCCL_DEBUG=INFO log (ring / tree over xGMI or a
#      fallback to PCIe / sockets is visible there), per-rank step times
set -u
N=${1:-8}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0   # dmabuf IPC only on this driver; RCCL needs it
export SARA_BENCH_STRICT_RCCL=1       # a communicator that cannot be formed stops the run here (bench.py alone falls back to torch.distributed and says so)
OUT=$R/gpurun_out/first_contact
mkdir -p "$OUT"
die() { echo "FIRST CONTACT FAILED at step $1: $2" | tee -a "$OUT/summary.txt"; exit 1; }
: > "$OUT/summary.txt"
port() { python - <<'PY'
import socket
s = socket.socket(); s.bind(("127.0.0.1", 0)); print(s.getsockname()[1])
PY
}

echo "== 0. node" | tee -a "$OUT/summary.txt"
NDEV=$(python -c "import torch; print(torch.cuda.device_count())")
echo "devices visible: $NDEV (asked for $N)" | tee -a "$OUT/summary.txt"
[ "$NDEV" -ge 2 ] || die 0 "fewer than 2 GPUs visible: nothing to find out here"
[ "$NDEV" -ge "$N" ] || { N=$NDEV; echo "running with N=$N" | tee -a "$OUT/summary.txt"; }
(rocm-smi --showtopo || true) > "$OUT/topology.txt" 2>&1
(rocm-smi --showproductname || true) >> "$OUT/topology.txt" 2>&1
python - <<'PY' | tee -a "$OUT/summary.txt"
import ctypes, glob, os, torch
print("torch", torch.__version__, "hip", torch.version.hip)
cands = [os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")] + \
        sorted(glob.glob("/opt/rocm/lib/librccl.so*"))
for p in cands:
    if os.path.exists(p):
        lib = ctypes.CDLL(p); v = ctypes.c_int(0); lib.ncclGetVersion(ctypes.byref(v))
        print("rccl", p, v.value)
PY

echo "== 1. build + one-GPU smoke" | tee -a "$OUT/summary.txt"
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > "$OUT/smoke.log" 2>&1 \
  || die 1 "see $OUT/smoke.log"
tail -1 "$OUT/smoke.log" | tee -a "$OUT/summary.txt"

echo "== 2. exchange logic over the loopback transport (one device)" | tee -a "$OUT/summary.txt"
timeout 900 python -m pytest tests/test_gpu_multi.py -m gpu -x -q > "$OUT/loopback_tests.log" 2>&1 \
  || die 2 "see $OUT/loopback_tests.log"
tail -1 "$OUT/loopback_tests.log" | tee -a "$OUT/summary.txt"

echo "== 3. N = 2 over RCCL, bytes verified at the root, both launch forms" | tee -a "$OUT/summary.txt"
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
  --master-addr 127.0.0.1 --master-port "$(port)" bench.py --gpus 2 --steps 5 --warmup 2 \
  --strict-h2h > "$OUT/n2_procs.json" 2> "$OUT/n2_procs.err" \
  || die 3 "one process per GPU: see $OUT/n2_procs.err"
timeout 900 python bench.py --gpus 2 --launch group --steps 5 --warmup 2 --strict-h2h \
  > "$OUT/n2_group.json" 2> "$OUT/n2_group.err" \
  || die 3 "one process driving 2 GPUs (ncclCommInitAll): see $OUT/n2_group.err"
python - "$OUT/n2_procs.json" "$OUT/n2_group.json" <<'PY' | tee -a "$OUT/summary.txt" || die 3 "a line is missing its RCCL evidence"
import json, sys
for p in sys.argv[1:]:
    d = json.loads([l for l in open(p) if l.startswith("{")][-1])
    print("%-40s transport=%s rccl_nranks=%s rccl_version=%s gather_verified=%s value=%.3g" % (
        p.split("/")[-1], d.get("transport"), d.get("rccl_nranks"), d.get("rccl_version"),
        d.get("gather_verified"), d["value"]))
    assert d.get("transport") == "rccl" and d.get("rccl_nranks") == 2, "not RCCL with 2 ranks"
    assert d.get("gather_verified"), "the gathered bytes were not verified"
PY

echo "== 4. N = 1, 2, 4, ... back to back (the driver's SCALE protocol)" | tee -a "$OUT/summary.txt"
: > "$OUT/scale.jsonl"
for n in 1 2 4 8; do
  [ "$n" -le "$N" ] || continue
  if [ "$n" -eq 1 ]; then
    timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 3 > "$OUT/n$n.json" 2> "$OUT/n$n.err" \
      || die 4 "N=1: see $OUT/n1.err"
  else
    timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node "$n" \
      --master-addr 127.0.0.1 --master-port "$(port)" bench.py --gpus "$n" --steps 20 --warmup 3 \
      --strict-h2h > "$OUT/n$n.json" 2> "$OUT/n$n.err" || die 4 "N=$n: see $OUT/n$n.err"
  fi
  grep '^{' "$OUT/n$n.json" | tail -1 >> "$OUT/scale.jsonl"
done
python - "$OUT/scale.jsonl" <<'PY' | tee -a "$OUT/summary.txt"
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1])]
base = rows[0]["value"]
print("%3s %14s %9s %10s %10s %8s %16s" % ("N", "keypoints/s", "x N=1", "ms/step", "transport", "verified", "host->host gray8"))
for d in rows:
    print("%3d %14.4g %9.2f %10.3f %10s %8s %16s" % (
        d["n_gpus"], d["value"], d["value"] / base, d["ms_per_step"], d.get("transport", "-"),
        d.get("gather_verified", "-"), "%.4g" % d["value_host_to_host_gray8"]
        if d.get("value_host_to_host_gray8") else "-"))
print("north_star target: >= 6x at 8 GPUs (weak scaling: 64 frames per GPU at every N)")
PY

echo "== 5. what RCCL chose at N = $N" | tee -a "$OUT/summary.txt"
NCCL_DEBUG=INFO NCCL_DEBUG_SUBSYS=INIT,GRAPH timeout 900 python -m torch.distributed.run --nnodes=1 \
  --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port "$(port)" bench.py --gpus "$N" \
  --steps 5 --warmup 2 --no-extras > "$OUT/nccl_debug.json" 2> "$OUT/nccl_debug.log" \
  || die 5 "see $OUT/nccl_debug.log"
grep -E "NCCL INFO (Channel|Ring|Trees|Connected|comm .* nranks|Using network|.*XGMI|.*P2P)" "$OUT/nccl_debug.log" \
  | sed 's/^.*NCCL INFO/NCCL INFO/' | sort | uniq -c | sort -rn | head -40 > "$OUT/nccl_choices.txt"
head -15 "$OUT/nccl_choices.txt" | tee -a "$OUT/summary.txt"
echo "FIRST CONTACT OK: $OUT/summary.txt, scale.jsonl" | tee -a "$OUT/summary.txt"
