"""Exhaustive host check of the orientation kernel's bin computation: for every
float in [0, float(2 pi)] the estimate-and-correct form against the reference
expression int(floor(double(a / float(2 pi) * 36))) (Orientation.hpp:118-119)."""
import numpy as np
TWO_PI = np.float32(2 * np.pi)
def exact(a):
    return np.floor((a / TWO_PI * np.float32(36)).astype(np.float64)).astype(np.int64)
def thresholds():
    thr = np.full(40, np.inf, np.float32)
    top = np.array([0x40c91000], np.uint32).view(np.float32)
    for k in range(40):
        lo, hi = 0, 0x40c91000
        if exact(top)[0] < k:
            continue
        while lo < hi:
            mid = (lo + hi) // 2
            if exact(np.array([mid], np.uint32).view(np.float32))[0] >= k:
                hi = mid
            else:
                lo = mid + 1
        thr[k] = np.array([lo], np.uint32).view(np.float32)[0]
    return thr
def fast(a, thr):
    kb = (a * np.float32(36 / (2 * np.pi))).astype(np.int32)
    kb = np.clip(kb, 0, 36)
    kb = kb + (a >= thr[kb + 1]).astype(np.int32) - (a < thr[kb]).astype(np.int32)
    return np.where(kb == 36, 0, kb)
if __name__ == "__main__":
    thr = thresholds()
    end = 0x40c90fdb + 1  # float(2 pi) inclusive
    bad = 0
    step = 1 << 24
    for s in range(0, end, step):
        a = np.arange(s, min(s + step, end), dtype=np.uint32).view(np.float32)
        bad += int(np.count_nonzero(fast(a, thr) != exact(a) % 36))
    print("floats checked:", end, "mismatches:", bad)
