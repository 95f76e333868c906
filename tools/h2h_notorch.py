"""The host -> host step (pinned float32 / gray8 frames ->
keypoints in pinned memory) measured WITHOUT torch in the process.

Why it exists: the torch wheel carries its own HIP runtime (ROCm 7.0.2 next to
the image's 7.2.0) and whichever libamdhip64.so.7 is loaded first serves the
whole process.  Under the 7.0.2 runtime an upload and a read-back on two
streams run one after the other on the copy engines; under 7.2.0 they overlap
(tools/ubench/pcie_duplex.hip).  A C++ caller of the library (Sara) links the
system runtime: this script is that case.  bench.py runs it as a subprocess
(--json) next to its own in-process measurement.

   python tools/h2h_notorch.py [--json] [--steps N] [--torch] [--kinds f32,u8]
"""
import argparse
import json
import os
import sys
import time

ap = argparse.ArgumentParser()
ap.add_argument("--json", action="store_true")
ap.add_argument("--steps", type=int, default=12)
ap.add_argument("--torch", action="store_true", help="import torch first (A/B)")
ap.add_argument("--kinds", default="f32,u8")
ap.add_argument("--frames", type=int, default=64)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--octaves", type=int, default=4)
ap.add_argument("--unique", type=int, default=16)
ap.add_argument("--torch-pinned", action="store_true")
ap.add_argument("--stage-first-first", action="store_true",
                help="measure the stage-first order before the submit-first one")
ap.add_argument("--registered", action="store_true",
                help="frames in hipHostRegister'ed memory instead of hipHostMalloc")
args = ap.parse_args()
if args.torch:
    import torch
    torch.cuda.init()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import sara_amd  # noqa: E402
from sara_amd import capi  # noqa: E402
from sara_amd.synth import synth_batch  # noqa: E402

B, W, H = args.frames, args.width, args.height
src = synth_batch(W, H, B, unique=min(args.unique, B))
lib = capi.load()
if args.registered:  # A/B: numpy's own pages, hipHostRegister'ed
    f = np.ascontiguousarray(src)
    u8 = np.ascontiguousarray(np.round(f * 255).astype(np.uint8))
    capi.check(lib.sara_hip_host_register(f.ctypes.data, f.nbytes))
    capi.check(lib.sara_hip_host_register(u8.ctypes.data, u8.nbytes))
elif args.torch_pinned:  # A/B: torch's caching host allocator
    tf = torch.from_numpy(np.ascontiguousarray(src)).pin_memory()
    tu = torch.from_numpy(np.round(src * 255).astype(np.uint8)).pin_memory()
    f, u8 = tf.numpy(), tu.numpy()
else:  # hipHostMalloc (sara_hip_host_alloc)
    f = sara_amd.pinned_empty(src.shape, np.float32)
    f[...] = src
    u8 = sara_amd.pinned_empty(src.shape, np.uint8)
    u8[...] = np.round(src * 255).astype(np.uint8)
ctx = sara_amd.SiftContext(W, H, B, sara_amd.ImagePyramidParams(
    0, 6, num_octaves_max=args.octaves))


def run(kind, stage_first, n):
    """-> (ms per step in the steady state, keypoints per step).  stage_first:
    stage(i + 1); collect(i - 1); submit_staged(i + 1) instead of submit(i + 1);
    collect(i)."""
    ptr, ch = (u8.ctypes.data, 1) if kind == "u8" else (f.ctypes.data, 0)
    tickets, kp = [], 0
    t0 = None
    for i in range(n + 4):
        if i == 4:
            t0, kp = time.perf_counter(), 0
        if stage_first:
            ctx.stage_raw(ptr, ch, B, W, H)
            if len(tickets) == 2:
                kp += int(ctx.collect(tickets.pop(0))[0][-1])
            tickets.append(ctx.submit_staged())
        else:
            tickets.append(ctx.submit_raw(ptr, ch, B, W, H))
            if len(tickets) == 2:
                kp += int(ctx.collect(tickets.pop(0))[0][-1])
    dt = time.perf_counter() - t0  # n submits and n collects: the steady state
    for t in tickets:
        ctx.collect(t)
    return 1e3 * dt / n, kp / n


out = {"runtime": [l.split()[-1] for l in open("/proc/self/maps")
                   if "libamdhip64" in l][:1]}
for kind in args.kinds.split(","):
    name = {"f32": "float32", "u8": "gray8"}[kind]
    if args.stage_first_first:
        b, kpb = run(kind, True, args.steps)
        a, kpa = run(kind, False, args.steps)
    else:
        a, kpa = run(kind, False, args.steps)
        b, kpb = run(kind, True, args.steps)
    out[name] = {"ms_per_step": min(a, b), "keypoints_per_s": 1e3 * kpa / min(a, b),
                 "ms_submit_then_collect": a, "ms_stage_collect_submit_staged": b}
    if not args.json:
        print("%-7s submit; collect %.2f ms   stage; collect; submit_staged %.2f ms "
              "per %d-frame step (%.0f keypoints)" % (name, a, b, B, kpa), flush=True)
if args.json:
    print(json.dumps(out))
else:
    print("HIP runtime:", out["runtime"])
