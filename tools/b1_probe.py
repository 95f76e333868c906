"""Developer probe: the drop-in's real operating point, ONE 1080p frame per
call (OdometryPipeline::detect_keypoints, SfM/Odometry/OdometryPipeline.cpp:
82-90).  Times host frame -> host keypoints through the Python mirror with a
persistent context, and the pieces (detect, counts, fetch)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import sara_amd  # noqa: E402
from sara_amd.synth import synth  # noqa: E402

W, H = 1920, 1080
img = synth(W, H, 1234)
p = sara_amd.ImagePyramidParams(0, 6, num_octaves_max=4)
ctx = sara_amd.SiftContext(W, H, 1, p)


def timeit(fn, n=50, warm=5):
    for _ in range(warm):
        fn()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    return 1e3 * (time.perf_counter() - t) / n


def det():
    ctx.detect(img)
    ctx.synchronize()


def det_counts():
    ctx.detect(img)
    return ctx.counts()


def det_fetch():
    ctx.detect(img)
    return ctx.fetch()


print("detect+sync        %.3f ms" % timeit(det))
print("detect+counts      %.3f ms" % timeit(det_counts))
print("detect+fetch       %.3f ms" % timeit(det_fetch))
print("keypoints", det_counts()[1])
if hasattr(sara_amd, "compute_sift_keypoints"):
    f = lambda: sara_amd.compute_sift_keypoints(img, p)
    print("compute_sift_keypoints %.3f ms" % timeit(f, n=20, warm=3))


def sub_col():
    return ctx.collect(ctx.submit(img))


u8 = (img * 255).astype(np.uint8)


def sub_col_u8():
    return ctx.collect(ctx.submit(u8))


print("submit+collect (float host frame)  %.3f ms" % timeit(sub_col))
print("submit+collect (gray8 host frame)  %.3f ms" % timeit(sub_col_u8))
