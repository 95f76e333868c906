"""Developer probe: where a pipelined host-to-host step (submit / collect) of
64 x 1080p spends its host time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import sara_amd
from sara_amd.synth import synth_batch
B, W, H = 64, 1920, 1080
f = synth_batch(W, H, B, unique=16)
u8 = torch.from_numpy(np.round(f * 255).astype(np.uint8)).pin_memory()
ctx = sara_amd.SiftContext(W, H, B, sara_amd.ImagePyramidParams(0, 6, num_octaves_max=4))
d = torch.from_numpy(f).cuda()
def run(kind):
    ts, tc = [], []
    t_prev = None
    t0 = time.perf_counter()
    n = 12
    for i in range(n):
        a = time.perf_counter()
        if kind == "u8host":
            t = ctx.submit_raw(u8.data_ptr(), 1, B, W, H)
        else:
            t = ctx.submit_raw(d.data_ptr(), 0, B, W, H, on_device=True)
        b = time.perf_counter()
        if t_prev is not None:
            ctx.collect(t_prev, with_descriptors=(kind != "nodesc"))
        c = time.perf_counter()
        ts.append(b - a); tc.append(c - b)
        t_prev = t
    ctx.collect(t_prev)
    dt = (time.perf_counter() - t0) / n
    print("%-8s step %.2f ms  submit %.2f ms  collect %.2f ms" % (kind, 1e3 * dt, 1e3 * np.mean(ts[2:]), 1e3 * np.mean(tc[2:])))
for k in ("u8host", "device", "nodesc", "u8host"):
    run(k)
def plain():
    ctx.detect_device(d.data_ptr(), B, W, H); ctx.counts()
for _ in range(3): plain()
t0 = time.perf_counter()
for _ in range(10): plain()
print("detect_device+counts %.2f ms" % (1e2 * (time.perf_counter() - t0)))
