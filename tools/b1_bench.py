"""Developer probe: the numbers bench.py reports as config2 / single_image (one
1080p frame per call, HIP-graph replay), without the rest of the benchmark.
   python tools/b1_bench.py [repeats [width height]]"""
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


W = int(sys.argv[2]) if len(sys.argv) > 3 else 1920
H = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
dev = torch.device("cuda:0")
one = synth_batch(W, H, 1)
d_one = torch.from_numpy(one).to(dev)
p4 = sara_amd.ImagePyramidParams(0, 6, num_octaves_max=4)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
with sara_amd.SiftContext(W, H, 1, p4, device=0) as c1:
    def run(stage):
        c1.detect_device(d_one.data_ptr(), 1, W, H, last_stage=stage)
        c1.synchronize()
    def enq(stage):
        t = time.perf_counter()
        c1.detect_device(d_one.data_ptr(), 1, W, H, last_stage=stage)
        dt = time.perf_counter() - t
        c1.synchronize()
        return dt
    for _ in range(reps):
        for st in (2, 5):
            for _ in range(20):
                enq(st)
            e = [enq(st) for _ in range(200)]
            print("stage %d: host time inside detect() %.4f ms (median %.4f)"
                  % (st, 1e3 * np.mean(e), 1e3 * np.median(e)))
        t2 = timed(lambda: run(2), 200, 20)
        t5 = timed(lambda: run(5), 200, 20)
        h2h = timed(lambda: c1.collect(c1.submit(one)), 100, 10)
        u8 = np.round(one * 255.0).astype(np.uint8)
        h2h8 = timed(lambda: c1.collect(c1.submit(u8)), 100, 10)
        print("config2 %.4f ms   full %.4f ms   h2h f32 %.4f ms   h2h u8 %.4f ms"
              % (1e3 * t2, 1e3 * t5, 1e3 * h2h, 1e3 * h2h8), flush=True)
