"""Developer probe: N host threads, one context each, calling detect() on a
device-resident 1080p frame (the per-camera call pattern of
OdometryPipeline::detect_keypoints): wall time / total calls.
   python tools/thread_calls.py [threads [calls]]"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import sara_amd  # noqa: E402
from sara_amd.synth import synth_batch  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 4
CALLS = int(sys.argv[2]) if len(sys.argv) > 2 else 200
W, H = 1920, 1080
frames = synth_batch(W, H, T, first_index=500)
p = sara_amd.ImagePyramidParams(0, 6, num_octaves_max=4)
ctxs = [sara_amd.SiftContext(W, H, 1, p) for _ in range(T)]
devs = [sara_amd.DeviceArray(frames[i:i + 1]) for i in range(T)]
start = threading.Barrier(T + 1)


def work(k):
    c = ctxs[k]
    for _ in range(5):
        c.detect_device(devs[k].ptr, 1, W, H)
        c.synchronize()
    start.wait()
    for _ in range(CALLS):
        c.detect_device(devs[k].ptr, 1, W, H)
        c.synchronize()
    start.wait()


ts = [threading.Thread(target=work, args=(k,)) for k in range(T)]
for t in ts:
    t.start()
start.wait()
t0 = time.perf_counter()
start.wait()
dt = time.perf_counter() - t0
for t in ts:
    t.join()
print("%d thread(s): %.3f ms per call (wall / total calls), %.3f ms per call per thread"
      % (T, 1e3 * dt / (T * CALLS), 1e3 * dt / CALLS))
