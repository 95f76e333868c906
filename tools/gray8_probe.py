"""Developer probe: 64 x 1080p gray8 frames resident in HBM through
sara_hip_sift_detect_u8, with (SARA_HIP_FUSE_GRAY8=1, default) and without the
first blur reading the bytes itself."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sara_amd
from sara_amd.synth import synth_batch
B, W, H = 64, 1920, 1080
f = synth_batch(W, H, B, unique=16)
u8 = torch.from_numpy(np.round(f * 255).astype(np.uint8)).cuda()
ctx = sara_amd.SiftContext(W, H, B, sara_amd.ImagePyramidParams(0, 6, num_octaves_max=4))
lib = sara_amd.capi.load()
ctx.batch = B  # the C entry point is called directly below
def step():
    sara_amd.capi.check(lib.sara_hip_sift_detect_u8(ctx._h, u8.data_ptr(), 0, 1, B, W, H, 1, 5, None))
    return ctx.counts()[1]
for _ in range(3):
    n = step()
t = time.perf_counter()
for _ in range(10):
    step()
print("fuse", os.environ.get("SARA_HIP_FUSE_GRAY8", "1"), "resident gray8: %.3f ms/step, %d keypoints"
      % ((time.perf_counter() - t) * 100, n))
ctx.close()
