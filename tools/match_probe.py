"""Timing probe for the matcher: the keypoints of two views of one 1080p scene
(about 4.3 k each), descriptors in HBM; both producers of the neighbour lists
(SARA_HIP_MATCH is read once per process, so the other one runs in a child)."""
import ctypes as C
import os
import subprocess
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sara_amd  # noqa: E402
from sara_amd import capi  # noqa: E402
from sara_amd.synth import synth  # noqa: E402

W, H = 1920, 1080
p = sara_amd.ImagePyramidParams(0, 6, num_octaves_max=4)
scene = synth(W + 24, H + 8, 1234)
ka = sara_amd.compute_sift_keypoints(np.ascontiguousarray(scene[:H, :W]), p)
kb = sara_amd.compute_sift_keypoints(np.ascontiguousarray(scene[8:, 24:]), p)
d1, d2 = ka.descriptor_matrix, kb.descriptor_matrix
n1, n2 = len(d1), len(d2)
lib = capi.load()
cap = 64 * (n1 + n2)
out = np.zeros(cap, capi.MATCH_DTYPE)
cnt = C.c_int()
t1 = torch.from_numpy(d1).cuda()
t2 = torch.from_numpy(d2).cuda()
mode = os.environ.get("SARA_HIP_MATCH", "default (mfma)")
for ratio in (0.6, 1.2):
    for name, a, b, dev in (("host pointers", d1.ctypes.data, d2.ctypes.data, 0),
                            ("device pointers", t1.data_ptr(), t2.data_ptr(), 1)):
        for _ in range(5):
            capi.check(lib.sara_hip_match_descriptors(a, n1, b, n2, 128, ratio, dev,
                                                      out.ctypes.data, cap,
                                                      C.byref(cnt), 0))
        reps = 50
        t0 = time.perf_counter()
        for _ in range(reps):
            capi.check(lib.sara_hip_match_descriptors(a, n1, b, n2, 128, ratio, dev,
                                                      out.ctypes.data, cap,
                                                      C.byref(cnt), 0))
        dt = (time.perf_counter() - t0) / reps
        print(f"[{mode}] ratio {ratio} {name}: {n1} x {n2}, {cnt.value} matches, "
              f"{dt*1e3:.3f} ms per call", flush=True)
# the batched entry point: P pairs per call (descriptors in HBM), per-pair time and
# the share of the bf16 MFMA peak the three products of every pair amount to
if "SARA_HIP_MATCH" not in os.environ:
    flop = 3 * 2.0 * n1 * n2 * 128          # hi*hi + hi*lo + lo*hi
    for P in (1, 2, 4, 8, 16, 32):
        arr = (capi.MatchPairStruct * P)()
        for k in range(P):
            arr[k] = capi.MatchPairStruct(t1.data_ptr(), t2.data_ptr(), n1, n2)
        offs = (C.c_int * (P + 1))()
        bout = np.zeros(P * (n1 + n2), capi.MATCH_DTYPE)
        for _ in range(3):
            capi.check(lib.sara_hip_match_descriptors_batch(
                arr, P, 128, 0.6, 1, bout.ctypes.data, len(bout), offs, 0))
        reps = max(4, 64 // P)
        t0 = time.perf_counter()
        for _ in range(reps):
            capi.check(lib.sara_hip_match_descriptors_batch(
                arr, P, 128, 0.6, 1, bout.ctypes.data, len(bout), offs, 0))
        dt = (time.perf_counter() - t0) / reps / P
        print(f"[batch] ratio 0.6 device pointers, P = {P:2d}: {dt*1e3:.3f} ms per pair, "
              f"{flop / dt / 1e12:.0f} TFLOP/s = {flop / dt / 2.5e15:.3f} of the bf16 "
              f"MFMA peak ({offs[P] // P} matches per pair)", flush=True)
if "SARA_HIP_MATCH" not in os.environ and "--no-child" not in sys.argv:
    subprocess.run([sys.executable, __file__, "--no-child"],
                   env=dict(os.environ, SARA_HIP_MATCH="exhaustive"))
