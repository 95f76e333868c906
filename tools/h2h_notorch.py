"""Developer probe: the host-to-host step (pinned float32 / gray8 frames ->
keypoints in pinned memory, two batches in flight) WITHOUT torch in the
process, and with it (`python tools/h2h_notorch.py torch`): torch's wheel
carries its own HIP runtime, and whichever libamdhip64.so.7 is loaded first
serves both."""
import os
import sys
import time

if len(sys.argv) > 1 and sys.argv[1] == "torch":
    import torch
    torch.cuda.init()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import sara_amd  # noqa: E402
from sara_amd import capi  # noqa: E402
from sara_amd.synth import synth_batch  # noqa: E402

B, W, H = 64, 1920, 1080
f = np.ascontiguousarray(synth_batch(W, H, B, unique=16))
u8 = np.ascontiguousarray(np.round(f * 255).astype(np.uint8))
lib = capi.load()
capi.check(lib.sara_hip_host_register(f.ctypes.data, f.nbytes))
capi.check(lib.sara_hip_host_register(u8.ctypes.data, u8.nbytes))
ctx = sara_amd.SiftContext(W, H, B, sara_amd.ImagePyramidParams(0, 6, num_octaves_max=4))


def run_staged(kind, n=12):
    """stage(i + 1); collect(i - 1); submit_staged(i + 1)"""
    ptr, ch = (u8.ctypes.data, 1) if kind == "u8" else (f.ctypes.data, 0)
    tickets = []
    ctx.stage_raw(ptr, ch, B, W, H)
    tickets.append(ctx.submit_staged())
    ts = [0.0, 0.0, 0.0]
    for i in range(n + 3):
        if i == 3:
            t0 = time.perf_counter()
            ts = [0.0, 0.0, 0.0]
        a = time.perf_counter()
        ctx.stage_raw(ptr, ch, B, W, H)
        b = time.perf_counter()
        if len(tickets) == 2:
            ctx.collect(tickets.pop(0))
        c = time.perf_counter()
        tickets.append(ctx.submit_staged())
        d = time.perf_counter()
        ts[0] += b - a
        ts[1] += c - b
        ts[2] += d - c
    for t in tickets:
        ctx.collect(t)
    print("%-4s host -> host, stage first: %.2f ms per 64-frame step  (host: stage %.2f collect %.2f submit_staged %.2f)" %
          (kind, 1e3 * (time.perf_counter() - t0) / n, 1e3 * ts[0] / n, 1e3 * ts[1] / n, 1e3 * ts[2] / n), flush=True)


def run(kind, n=12):
    t_prev = None
    ts = [0.0, 0.0]
    for i in range(n + 3):
        if i == 3:
            t0 = time.perf_counter()
            ts = [0.0, 0.0]
        a = time.perf_counter()
        if kind == "u8":
            t = ctx.submit_raw(u8.ctypes.data, 1, B, W, H)
        else:
            t = ctx.submit_raw(f.ctypes.data, 0, B, W, H)
        b = time.perf_counter()
        if t_prev is not None:
            ctx.collect(t_prev)
        ts[0] += b - a
        ts[1] += time.perf_counter() - b
        t_prev = t
    ctx.collect(t_prev)
    print("%-4s host -> host: %.2f ms per 64-frame step  (host: submit %.2f collect %.2f)" %
          (kind, 1e3 * (time.perf_counter() - t0) / n, 1e3 * ts[0] / n, 1e3 * ts[1] / n), flush=True)


for k in (("f32", "f32") if "f32only" in sys.argv else ("f32", "u8", "f32", "u8")):
    run(k)
    run_staged(k)
import ctypes
print("HIP runtime:", [l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l][:1])
