import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import sara_amd
from sara_amd import capi
from sara_amd.synth import synth
W, H = 1920, 1080
p = sara_amd.ImagePyramidParams(0, 6, num_octaves_max=4)
scene = synth(W + 24, H + 8, 1234)
ka = sara_amd.compute_sift_keypoints(np.ascontiguousarray(scene[:H, :W]), p)
kb = sara_amd.compute_sift_keypoints(np.ascontiguousarray(scene[8:, 24:]), p)
d1, d2 = ka.descriptor_matrix, kb.descriptor_matrix
RATIO = float(sys.argv[1]) if len(sys.argv) > 1 else 0.6
with sara_amd.DeviceArray(d1) as t1, sara_amd.DeviceArray(d2) as t2:
    lib = capi.load(); out = np.zeros(2*(len(d1)+len(d2)), capi.MATCH_DTYPE); cnt = C.c_int()
    for i in range(4):
        if i == 3: print("---- last call", file=sys.stderr, flush=True)
        capi.check(lib.sara_hip_match_descriptors(t1.ptr, len(d1), t2.ptr, len(d2), 128, RATIO, 1, out.ctypes.data, len(out), C.byref(cnt), 0))
