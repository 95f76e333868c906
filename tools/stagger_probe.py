"""Probe: two 32-frame contexts whose steps are offset by half a step, so that
one half-batch's per-keypoint kernels (latency-bound) run next to the other's
pyramid / scan / gradients (memory-bound).  Compared with one 64-frame context."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import sara_amd
from sara_amd.synth import synth_batch

W, H, B = 1920, 1080, 64
frames = torch.from_numpy(synth_batch(W, H, B, unique=16)).cuda()
params = sara_amd.ImagePyramidParams(0, 6, 2 ** (1 / 3), 1, 0.5, 1.6, 4)

def one(steps=10):
    c = sara_amd.SiftContext(W, H, B, params)
    for _ in range(3):
        c.detect_device(frames.data_ptr(), B, W, H); c.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        c.detect_device(frames.data_ptr(), B, W, H); c.synchronize()
    dt = (time.perf_counter() - t0) / steps
    c.close()
    return dt * 1e3

def staggered(offset_ms, steps=10):
    per = B // 2
    a = sara_amd.SiftContext(W, H, per, params)
    b = sara_amd.SiftContext(W, H, per, params)
    fa, fb = frames[:per], frames[per:]
    for _ in range(2):
        a.detect_device(fa.data_ptr(), per, W, H); b.detect_device(fb.data_ptr(), per, W, H)
        a.synchronize(); b.synchronize()
    # A starts; B follows offset_ms later; from then on each restarts as soon as it is done
    t0 = time.perf_counter()
    a.detect_device(fa.data_ptr(), per, W, H)
    time.sleep(offset_ms * 1e-3)
    b.detect_device(fb.data_ptr(), per, W, H)
    for _ in range(steps - 1):
        a.synchronize(); a.detect_device(fa.data_ptr(), per, W, H)
        b.synchronize(); b.detect_device(fb.data_ptr(), per, W, H)
    a.synchronize(); b.synchronize()
    dt = (time.perf_counter() - t0) / steps
    a.close(); b.close()
    return dt * 1e3

print("one 64-frame context: %.3f ms/step" % one())
for off in (0.0, 1.0, 2.0, 3.0):
    print("2 x 32 frames, offset %.1f ms: %.3f ms per 64 frames" % (off, staggered(off)))
print("one 64-frame context: %.3f ms/step" % one())
