"""Deterministic synthetic frames for the benchmark and the parity tests
(SURVEY.md section 8d): float32 in [0,1] =
clamp(0.5 + sum of Gaussian blobs + 0.02 * band-limited noise).

n = w*h/400 blobs, centres uniform, radius rho log-uniform in [1.2, 16] px
(quantised to 24 levels so that each level is one separable Gaussian filter of
an impulse image), amplitude uniform in [0.15, 0.5] with random sign, noise =
N(0,1) blurred with sigma 1.  RNG = numpy PCG64 seeded ``1234 + frame_index``.
"""
import numpy as np
from scipy import ndimage

BASE_SEED = 1234


def synth(width, height, seed=BASE_SEED):
    rng = np.random.default_rng(int(seed))
    n = max(1, (width * height) // 400)
    cx = rng.integers(0, width, size=n)
    cy = rng.integers(0, height, size=n)
    levels = np.exp(np.linspace(np.log(1.2), np.log(16.0), 24))
    lvl = rng.integers(0, len(levels), size=n)
    amp = rng.uniform(0.15, 0.5, size=n) * rng.choice([-1.0, 1.0], size=n)
    img = np.zeros((height, width), np.float64)
    for li, rho in enumerate(levels):
        m = lvl == li
        if not m.any():
            continue
        imp = np.zeros((height, width), np.float64)
        # unit-peak blob = impulse * (2 pi rho^2) filtered by a normalised Gaussian
        np.add.at(imp, (cy[m], cx[m]), amp[m] * (2.0 * np.pi * rho * rho))
        img += ndimage.gaussian_filter(imp, rho, mode="constant", truncate=4.0)
    noise = ndimage.gaussian_filter(rng.standard_normal((height, width)), 1.0,
                                    mode="nearest")
    noise /= max(noise.std(), 1e-12)
    out = 0.5 + img + 0.02 * noise
    return np.clip(out, 0.0, 1.0).astype(np.float32)


def synth_batch(width, height, count, first_index=0, unique=None):
    """``count`` frames with seeds BASE_SEED + first_index + i.  When ``unique``
    is given, only that many frames are generated and the rest are their
    horizontal / vertical flips (distinct images, same statistics), which
    keeps the host-side set-up time of the benchmark short."""
    unique = count if unique is None else max(1, min(unique, count))
    base = [synth(width, height, BASE_SEED + first_index + i)
            for i in range(unique)]
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
