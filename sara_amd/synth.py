"""Deterministic synthetic frames for the benchmark and the parity tests
(SURVEY.md section 8d): float32 in [0,1] =
clamp(0.5 + sum of Gaussian blobs + 0.02 * band-limited noise).

The generator itself is C++ (sara_amd/csrc/synth.cpp -> lib/libsara_synth.so,
declared in include/sara_synth.h): SplitMix64 seeded ``1234 + frame_index``,
n = w*h/400 blobs, centres uniform, radius log-uniform in [1.2, 16] px,
amplitude uniform in [0.15, 0.5] with random sign, noise = N(0,1) blurred with
sigma 1.  This module is the ctypes binding.
"""
import ctypes
import os

import numpy as np

BASE_SEED = 1234
_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib",
                            "libsara_synth.so")
        if not os.path.exists(path):
            raise RuntimeError(
                "%s is missing: run `make -C sara_amd/csrc` (or "
                "__graft_entry__.build())" % path)
        lib = ctypes.CDLL(path)
        lib.sara_synth_frame.argtypes = [ctypes.c_int, ctypes.c_int,
                                         ctypes.c_uint64, ctypes.c_void_p]
        lib.sara_synth_frame.restype = ctypes.c_int
        lib.sara_synth_batch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                         ctypes.c_uint64, ctypes.c_void_p,
                                         ctypes.c_int]
        lib.sara_synth_batch.restype = ctypes.c_int
        _LIB = lib
    return _LIB


def _threads():
    """CPUs this process may use (affinity capped by the cgroup quota)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def synth(width, height, seed=BASE_SEED):
    out = np.empty((height, width), np.float32)
    if _lib().sara_synth_frame(width, height, int(seed), out.ctypes.data) != 0:
        raise ValueError("bad frame size")
    return out


def synth_batch(width, height, count, first_index=0, unique=None):
    """``count`` frames with seeds BASE_SEED + first_index + i.  When ``unique``
    is given, only that many frames are generated and the rest are their
    horizontal / vertical flips (distinct images, same statistics)."""
    unique = count if unique is None else max(1, min(unique, count))
    base = np.empty((unique, height, width), np.float32)
    if _lib().sara_synth_batch(width, height, unique,
                               BASE_SEED + int(first_index), base.ctypes.data,
                               _threads()) != 0:
        raise ValueError("bad batch size")
    if unique == count:
        return base
    out = np.empty((count, height, width), np.float32)
    for i in range(count):
        f = base[i % unique]
        k = (i // unique) % 4
        if k == 1:
            f = f[:, ::-1]
        elif k == 2:
            f = f[::-1, :]
        elif k == 3:
            f = f[::-1, ::-1]
        out[i] = f
    return out
