"""Developer probe: how far is one 1080p frame per call from the floor its own
critical chain sets?  The chain of a call is octave 0's: base blur, five
incremental blurs, extremum scan, then finish_sites -> ordering -> orientations
-> peak scan -> descriptors.  A context with num_octaves_max = 1 runs exactly
that chain and nothing beside it (its lists are a little shorter: the keypoints
of octaves 1-3 are missing, 27 % of them): its time is a lower bound for any
schedule of the four-octave call with these kernels.
   python tools/b1_floor.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import sara_amd  # noqa: E402
from sara_amd.synth import synth_batch  # noqa: E402


def timed(fn, n, warm):
    for _ in range(warm):
        fn()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    return (time.perf_counter() - t) / n


W, H = 1920, 1080
dev = torch.device("cuda:0")
one = synth_batch(W, H, 1)
d_one = torch.from_numpy(one).to(dev)
print("orientation kernel: %s" % os.environ.get("SARA_HIP_ORI", "default"))
for noct in (1, 2, 3, 4):
    p = sara_amd.ImagePyramidParams(0, 6, num_octaves_max=noct)
    with sara_amd.SiftContext(W, H, 1, p, device=0) as c1:
        def run(stage):
            c1.detect_device(d_one.data_ptr(), 1, W, H, last_stage=stage)
            c1.synchronize()
        best2 = min(timed(lambda: run(2), 300, 30) for _ in range(3))
        best5 = min(timed(lambda: run(5), 300, 30) for _ in range(3))
        run(5)
        _, kp = c1.counts()
        print("octaves %d: pyramid + extrema %.4f ms   full SIFT %.4f ms   "
              "%d keypoints" % (noct, 1e3 * best2, 1e3 * best5, int(kp)), flush=True)
