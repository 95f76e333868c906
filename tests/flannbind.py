"""ctypes binding of oracle/_ref/libflann_ref.so: the reference's own vendored
FLANN (header-only, /root/reference/cpp/third-party/flann), compiled by
`make -C oracle _ref` and called as FeatureMatching/AnnMatcher.cpp calls it.

TEST INFRASTRUCTURE ONLY.  /root/reference does not exist on the GPU box: there
the prebuilt .so travels with the snapshot, and where neither exists the tests
fall back to the committed fixture tests/golden/flann_pins.npz."""
import ctypes as C
import os
import subprocess

import numpy as np

import refbind as rb

_REF_DIR = os.path.join(rb.ORACLE_DIR, "_ref")
_LIB_PATH = os.path.join(_REF_DIR, "libflann_ref.so")
FLANN_HEADER = "/root/reference/cpp/third-party/flann/src/cpp/flann/flann.hpp"

LINEAR, KDTREE8 = 0, 1
_lib = None


def available():
    """The library is there or can be built (reference present)."""
    return os.path.exists(_LIB_PATH) or os.path.exists(FLANN_HEADER)


def lib():
    global _lib
    if _lib is None:
        if os.path.exists(FLANN_HEADER):
            subprocess.check_call(["make", "-s", "-C", rb.ORACLE_DIR, "_ref"])
        _lib = C.CDLL(_LIB_PATH)
    return _lib


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def knn(data, queries, k, kind=LINEAR):
    d, q = _f(data), _f(queries)
    idx = np.zeros((len(q), k), np.int32)
    dist = np.zeros((len(q), k), np.float32)
    fn = lib().flann_ref_knn
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int,
                   C.c_int, C.c_void_p, C.c_void_p]
    assert fn(d.ctypes.data, len(d), d.shape[1], q.ctypes.data, len(q), k, kind,
              idx.ctypes.data, dist.ctypes.data) == 0
    return idx, dist


def radius(data, queries, radii, kind=LINEAR, max_nn=None):
    """-> list of (indices, distances) per query, in FLANN's returned order."""
    d, q = _f(data), _f(queries)
    r = _f(radii)
    max_nn = max_nn or len(d)
    idx = np.zeros((len(q), max_nn), np.int32)
    dist = np.zeros((len(q), max_nn), np.float32)
    count = np.zeros(len(q), np.int32)
    fn = lib().flann_ref_radius
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                   C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    assert fn(d.ctypes.data, len(d), d.shape[1], q.ctypes.data, len(q),
              r.ctypes.data, max_nn, kind, idx.ctypes.data, dist.ctypes.data,
              count.ctypes.data) == 0
    return [(idx[i, :count[i]].copy(), dist[i, :count[i]].copy())
            for i in range(len(q))]


def compute_matches(desc1, desc2, ratio, kind=LINEAR):
    a, b = _f(desc1), _f(desc2)
    fn = lib().flann_ref_compute_matches
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_float,
                   C.c_int, C.c_void_p, C.c_int]
    cap = 2 * (len(a) + len(b)) + 16
    while True:
        out = np.zeros(cap, rb.MATCH_DTYPE)
        n = fn(a.ctypes.data, len(a), b.ctypes.data, len(b), a.shape[1], ratio,
               kind, out.ctypes.data, cap)
        assert n >= 0
        if n <= cap:
            return out[:n]
        cap = n


def compute_self_matches(desc, regions, ratio=1.2, metric_thres=0.5,
                         pixel_thres=10.0, kind=LINEAR):
    a = _f(desc)
    f = np.ascontiguousarray(rb.match_features(regions))
    fn = lib().flann_ref_compute_self_matches
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float,
                   C.c_float, C.c_int, C.c_void_p, C.c_int]
    cap = 4 * len(a) + 16
    while True:
        out = np.zeros(cap, rb.MATCH_DTYPE)
        n = fn(a.ctypes.data, f.ctypes.data, len(a), a.shape[1], ratio,
               metric_thres, pixel_thres, kind, out.ctypes.data, cap)
        assert n >= 0
        if n <= cap:
            return out[:n]
        cap = n
