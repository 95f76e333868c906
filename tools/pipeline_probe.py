"""Developer probe: 64 x 1080p steps on frames resident in HBM, synchronised
step by step (detect + counts, what bench.py times at N = 1) against two
batches in flight (submit batch i + 1, then the counts of batch i).
   python tools/pipeline_probe.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import sara_amd  # noqa: E402
from sara_amd.synth import synth_batch  # noqa: E402

W, H, B = 1920, 1080, 64
frames = torch.from_numpy(synth_batch(W, H, B, unique=8)).to("cuda:0")
p = sara_amd.ImagePyramidParams(0, 6, num_octaves_max=4)
with sara_amd.SiftContext(W, H, B, p, device=0) as c:
    def serial(n):
        kp = 0
        for _ in range(n):
            c.detect_device(frames.data_ptr(), B, W, H)
            kp += c.counts()[1]
        return kp

    def piped(n):
        kp, prev = 0, None
        for _ in range(n):
            t = c.submit_raw(frames.data_ptr(), 0, B, W, H, on_device=True)
            if prev is not None:
                kp += c.ticket_counts(prev)[1]
                c.collect_into(prev, None, None, None)
            prev = t
        kp += c.ticket_counts(prev)[1]
        c.collect_into(prev, None, None, None)
        return kp

    for name, fn in (("serial", serial), ("two in flight", piped)) * 3:
        fn(8)
        t = time.perf_counter()
        kp = fn(40)
        dt = (time.perf_counter() - t) / 40
        print("%-14s %.3f ms per step, %.2f M keypoints/s" % (name, 1e3 * dt, 1e-6 * kp / 40 / dt),
              flush=True)
