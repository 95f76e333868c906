"""Developer probe: one 1080p frame per call in a process WITHOUT torch, i.e. on
the system's ROCm runtime (tools/b1_bench.py and bench.py import torch and run
on the runtime it bundles).
   python tools/b1_notorch.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sara_amd  # noqa: E402
from sara_amd.synth import synth_batch  # noqa: E402

W, H = 1920, 1080
frame = synth_batch(W, H, 1)
p = sara_amd.ImagePyramidParams(0, 6, num_octaves_max=4)
with sara_amd.DeviceArray(frame) as d, sara_amd.SiftContext(W, H, 1, p, device=0) as c:
    for stage, name in ((2, "config 2 (pyramid + extrema)"), (5, "full SIFT")):
        def run():
            c.detect_device(d.ptr, 1, W, H, last_stage=stage)
            c.synchronize()
        for _ in range(30):
            run()
        best = 1e9
        for _ in range(5):
            t = time.perf_counter()
            for _ in range(300):
                run()
            best = min(best, (time.perf_counter() - t) / 300)
        print("%s: %.4f ms per call" % (name, 1e3 * best), flush=True)
assert "torch" not in sys.modules
