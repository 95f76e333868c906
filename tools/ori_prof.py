"""Development aid: per-phase wave cycles of the orientation kernel (library
built with -DSARA_ORI_PROF: tools/ab_build.sh oprof "-DSARA_ORI_PROF"
feature_kernels.hip; SARA_HIP_SIFT_LIB=sara_amd/lib/ab/lib_oprof.so python
tools/ori_prof.py)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import sara_amd
from sara_amd import capi
from sara_amd.synth import synth_batch
B, W, H = 64, 1920, 1080
frames = synth_batch(W, H, B, unique=16)
ctx = sara_amd.SiftContext(W, H, B, sara_amd.ImagePyramidParams(0, 6, num_octaves_max=4))
lib = capi.load()
out = (ctypes.c_ulonglong * 8)()
ctx.detect(frames); ctx.counts()
lib.sara_hip_debug_desc_prof(out, 1)
for _ in range(3):
    ctx.detect(frames); ctx.counts()
lib.sara_hip_debug_desc_prof(out, 0)
v = np.array(list(out), dtype=np.float64) / 3
names = ["-", "gather+bin+weight", "sort machinery", "replay loop", "smooth+peaks+write", "placement (whole-patch kernel)", "-", "item total"]
for n, x in zip(names, v):
    if n != "-":
        print("%-20s %14.0f  %5.1f %%" % (n, x, 100 * x / v[7]))
print("setup + rest        %14.0f  %5.1f %%" % (v[7] - v[1:6].sum(), 100 * (v[7] - v[1:6].sum()) / v[7]))
