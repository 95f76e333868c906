"""Timing probe for sara_hip_match_descriptors: the keypoints of two 1080p
synthetic frames (about 4.2 k each), host pointers and device pointers."""
import ctypes as C
import time
import numpy as np
import torch
import sara_amd
from sara_amd import capi
from sara_amd.synth import synth

p = sara_amd.ImagePyramidParams(0, 6, num_octaves_max=4)
ka = sara_amd.compute_sift_keypoints(synth(1920, 1080, 1234), p)
kb = sara_amd.compute_sift_keypoints(synth(1920, 1080, 1235), p)
d1, d2 = ka.descriptor_matrix, kb.descriptor_matrix
n1, n2 = len(d1), len(d2)
lib = capi.load()
out = np.zeros(n1 + n2, capi.MATCH_DTYPE)
cnt = C.c_int()
t1 = torch.from_numpy(d1).cuda(); t2 = torch.from_numpy(d2).cuda()
for name, a, b, dev in (("host pointers", d1.ctypes.data, d2.ctypes.data, 0),
                        ("device pointers", t1.data_ptr(), t2.data_ptr(), 1)):
    for _ in range(3):
        capi.check(lib.sara_hip_match_descriptors(a, n1, b, n2, 128, 0.6, dev,
                                                  out.ctypes.data, n1 + n2,
                                                  C.byref(cnt), 0))
    t0 = time.perf_counter()
    for _ in range(20):
        capi.check(lib.sara_hip_match_descriptors(a, n1, b, n2, 128, 0.6, dev,
                                                  out.ctypes.data, n1 + n2,
                                                  C.byref(cnt), 0))
    dt = (time.perf_counter() - t0) / 20
    pairs = 2.0 * n1 * n2
    print(f"{name}: {n1} x {n2} descriptors, {cnt.value} matches, {dt*1e3:.3f} ms per call, "
          f"{pairs * 128 * 3 / dt / 1e12:.2f} T flop/s equivalent", flush=True)
