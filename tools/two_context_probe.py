"""Developer probe: do the per-keypoint stages of one batch (orientation,
descriptors: latency- / LDS-bound) overlap with the pyramid and scan stages of
the next one (bandwidth- / issue-bound) when two contexts take turns?
   python tools/two_context_probe.py [frames per context]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import sara_amd  # noqa: E402
from sara_amd.synth import synth_batch  # noqa: E402

W, H = 1920, 1080
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
frames = torch.from_numpy(synth_batch(W, H, B, unique=8)).to(dev)
p = sara_amd.ImagePyramidParams(0, 6, num_octaves_max=4)
ctxs = [sara_amd.SiftContext(W, H, B, p, device=0) for _ in range(2)]


def run(n_ctx, steps):
    for k in range(steps):
        ctxs[k % n_ctx].detect_device(frames.data_ptr(), B, W, H)
    for c in ctxs[:n_ctx]:
        c.synchronize()


for n_ctx in (1, 2, 1, 2):
    run(n_ctx, 4)
    t = time.perf_counter()
    run(n_ctx, 20)
    dt = (time.perf_counter() - t) / 20
    print("%d context(s): %.3f ms per step of %d frames" % (n_ctx, 1e3 * dt, B), flush=True)
