"""Probe: one 64-frame context versus 2 x 32 / 4 x 16 frame contexts whose
pipelines run concurrently on their own streams (does the hardware overlap the
HBM-bound and the VALU-bound stages of different sub-batches?)."""
import sys, time
import numpy as np
import torch
import sara_amd
from sara_amd.synth import synth_batch

W, H, B = 1920, 1080, 64
frames = torch.from_numpy(synth_batch(W, H, B, unique=4)).cuda()
params = sara_amd.ImagePyramidParams(0, 6, 2 ** (1 / 3), 1, 0.5, 1.6, 4)


def bench(nsplit, steps=8):
    per = B // nsplit
    ctxs = [sara_amd.SiftContext(W, H, per, params) for _ in range(nsplit)]
    def step():
        for i, c in enumerate(ctxs):
            c.detect_device(frames[i * per:(i + 1) * per].data_ptr(), per, W, H)
        for c in ctxs:
            c.synchronize()
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    n = sum(c.counts()[1] for c in ctxs)
    for c in ctxs:
        c.close()
    return dt * 1e3, n

for ns in (1, 2, 4, 1, 2, 4):
    ms, n = bench(ns)
    print(f"split {ns}: {ms:.3f} ms/step  {n} keypoints  {n / ms / 1e3:.2f} M kp/s", flush=True)
