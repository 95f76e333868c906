"""Shared helpers of the parity tests."""
import hashlib
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def load_sunflower_gray():
    """tests/golden/sunflower_rgb8.png -> gray32f with the reference's
    Rgb8 -> float conversion (refbind.rgb8_to_gray32f)."""
    from PIL import Image
    import refbind as rb
    rgb = np.array(Image.open(os.path.join(GOLDEN, "sunflower_rgb8.png")))
    return rb.rgb8_to_gray32f(rgb)


def regions_from_bytes(b):
    import refbind as rb
    return np.ascontiguousarray(b).view(rb.OEREGION_DTYPE).reshape(-1)


def assert_regions_equal(a, b, atol_xy=0.0, atol_theta=0.0, rtol_shape=0.0):
    """Field-by-field comparison of two OERegion arrays (same order)."""
    assert len(a) == len(b), (len(a), len(b))
    assert np.array_equal(a["type"], b["type"])
    assert np.array_equal(a["extremum_type"], b["extremum_type"])
    if atol_xy == 0.0:
        assert np.array_equal(a["coords"], b["coords"])
        assert np.array_equal(a["extremum_value"], b["extremum_value"])
    else:
        assert np.allclose(a["coords"], b["coords"], rtol=0, atol=atol_xy)
        assert np.allclose(a["extremum_value"], b["extremum_value"], rtol=1e-6,
                           atol=1e-8)
    if rtol_shape == 0.0:
        assert np.array_equal(a["shape_matrix"], b["shape_matrix"])
    else:
        assert np.allclose(a["shape_matrix"], b["shape_matrix"],
                           rtol=rtol_shape, atol=0)
    if atol_theta == 0.0:
        assert np.array_equal(a["orientation"], b["orientation"])
    else:
        assert np.allclose(a["orientation"], b["orientation"], rtol=0,
                           atol=atol_theta)


def dot_grid(w, h, pitch=10, seed=3):
    """A calibration-target-like frame: 3 x 3 bright dots on a `pitch`-pixel
    grid, every dot moved by up to one pixel (seeded).  Far denser in blobs
    than any photograph: a 1920 x 1080 frame gives over 20 000 SIFT keypoints,
    above the default list capacity of a context (w * h / 128 = 16 200)."""
    rng = np.random.default_rng(seed)
    img = np.zeros((h, w), np.float32)
    ys = np.arange(pitch // 2, h - pitch // 2, pitch)
    xs = np.arange(pitch // 2, w - pitch // 2, pitch)
    jy = rng.integers(-1, 2, (len(ys), len(xs)))
    jx = rng.integers(-1, 2, (len(ys), len(xs)))
    for i, y in enumerate(ys):
        for j, x in enumerate(xs):
            yy, xx = y + jy[i, j], x + jx[i, j]
            img[yy - 1:yy + 2, xx - 1:xx + 2] = 1.0
    return img
