"""PCIe-inclusive rate of the benchmark workload (64 x 1080p, full SIFT): frames
start in pinned host memory and are uploaded by detect() itself, serially with
the compute (no double buffering).  Float frames, gray8 and RGB8 frames
(converted on the device).  Not the bench's `value`; quoted in DESIGN.md."""
import time
import numpy as np
import torch
import sara_amd
from sara_amd.synth import synth_batch

W, H, B = 1920, 1080, 64
f32 = synth_batch(W, H, B, unique=4)
g8 = np.clip(f32 * 255.0 + 0.5, 0, 255).astype(np.uint8)
rgb = np.repeat(g8[..., None], 3, axis=-1)


def pinned(a):
    t = torch.from_numpy(np.ascontiguousarray(a)).pin_memory()
    return t, t.numpy()

params = sara_amd.ImagePyramidParams(0, 6, num_octaves_max=4)
with sara_amd.SiftContext(W, H, B, params) as ctx:
    for name, arr, call in (("float32 frames (8.3 MB each)", f32, ctx.detect),
                            ("gray8 frames (2.1 MB each)", g8, ctx.detect_u8),
                            ("RGB8 frames (6.2 MB each)", rgb, ctx.detect_u8)):
        keep, host = pinned(arr)
        for _ in range(2):
            call(host)
            ctx.counts()
        t0 = time.perf_counter()
        n = 0
        steps = 8
        for _ in range(steps):
            call(host)
            n += ctx.counts()[1]
        dt = (time.perf_counter() - t0) / steps
        print(f"{name}: {dt * 1e3:.2f} ms per 64-frame step, {n / steps / dt / 1e6:.1f} M keypoints/s", flush=True)
        # double-buffered: stage(i+1) is issued right after detect_staged(i)
        ctx.stage(host)
        for _ in range(2):
            ctx.detect_staged()
            ctx.stage(host)
            ctx.counts()
        t0 = time.perf_counter()
        n = 0
        for _ in range(steps):
            ctx.detect_staged()
            ctx.stage(host)
            n += ctx.counts()[1]
        dt = (time.perf_counter() - t0) / steps
        ctx.detect_staged()
        ctx.counts()
        print(f"    double-buffered (stage / detect_staged): {dt * 1e3:.2f} ms per step, "
              f"{n / steps / dt / 1e6:.1f} M keypoints/s", flush=True)
