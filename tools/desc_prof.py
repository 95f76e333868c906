"""Development aid: per-phase wave cycles of the descriptor kernel (library
built with -DSARA_DESC_PROF, tools/ab_build.sh prof "-DSARA_DESC_PROF"
feature_kernels.hip).  SARA_HIP_SIFT_LIB=sara_amd/lib/ab/lib_prof.so python
tools/desc_prof.py"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import sara_amd  # noqa: E402
from sara_amd import capi  # noqa: E402
from sara_amd.synth import synth_batch  # noqa: E402

B, W, H = 64, 1920, 1080
frames = synth_batch(W, H, B, unique=16)
ctx = sara_amd.SiftContext(W, H, B, sara_amd.ImagePyramidParams(0, 6, num_octaves_max=4))
lib = capi.load()
out = (ctypes.c_ulonglong * 8)()
ctx.detect(frames)
ctx.counts()
lib.sara_hip_debug_desc_prof(out, 1)
for _ in range(3):
    ctx.detect(frames)
    counts, total = ctx.counts()
lib.sara_hip_debug_desc_prof(out, 0)
v = np.array(list(out), dtype=np.float64) / 3
names = ["setup", "trig+zero", "row table", "steps", "finalize", "nsteps", "tables", "item total"]
for n, x in zip(names, v):
    print("%-12s %14.0f" % (n, x))
print("keypoints/step", total, " cycles/item %.0f" % (v[7] / max(total, 1)),
      " cycles/step %.0f" % (v[3] / max(v[5], 1)), " steps/item %.1f" % (v[5] / max(total, 1)))
