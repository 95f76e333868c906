"""Developer probe: time per frame of full SIFT on 1080p frames resident in HBM
against the batch size of the call (where do the schedules hand over?).
   python tools/batch_sweep.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import sara_amd  # noqa: E402
from sara_amd.synth import synth_batch  # noqa: E402

W, H = 1920, 1080
dev = torch.device("cuda:0")
frames = torch.from_numpy(synth_batch(W, H, 64, unique=8)).to(dev)
p = sara_amd.ImagePyramidParams(0, 6, num_octaves_max=4)
BATCHES = [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4, 6, 8, 9, 12, 16, 24, 32, 48, 64]
for B in BATCHES:
    with sara_amd.SiftContext(W, H, B, p, device=0) as c:
        def run():
            c.detect_device(frames.data_ptr(), B, W, H)
            c.synchronize()
        for _ in range(5):
            run()
        best = 1e9
        for _ in range(3):
            t = time.perf_counter()
            for _ in range(20):
                run()
            best = min(best, (time.perf_counter() - t) / 20)
        _, kp = c.counts()
        print("B = %2d: %.3f ms per call, %.4f ms per frame, %.1f M keypoints/s"
              % (B, 1e3 * best, 1e3 * best / B, 1e-6 * float(kp) / best), flush=True)
