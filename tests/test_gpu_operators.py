"""-m gpu: operator-level parity.  The reference's own unit tests (SURVEY.md
section 4) restated against the HIP kernels through the C-ABI, plus
kernel-vs-oracle comparisons on seeded inputs.  Bit-exact unless stated.
"""
import math

import numpy as np
import pytest

import sara_amd

pytestmark = pytest.mark.gpu

RNG = np.random.default_rng(7)


# test_imageprocessing_linear_filtering.cpp:136-187 on the HIP blur.
@pytest.mark.parametrize("n,truncate", [(3, 1.0), (9, 4.0), (65, 4.0)])
def test_gaussian_on_dirac(n, truncate):
    src = np.zeros((n, n), np.float32)
    src[n // 2, n // 2] = 1
    out = sara_amd.apply_gaussian_filter(src, 1.0, truncate)
    c = n // 2
    i, j = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    true = np.exp(-((i - c) ** 2.0 + (j - c) ** 2.0) / 2.0)
    if n == 3:
        true = np.exp(-((i - c) ** 2.0 + (j - c) ** 2.0) / 2.0)
    true /= true.sum()
    assert np.linalg.norm(true - out) < 1e-5


def test_blur_of_constant_is_constant():
    src = np.full((37, 53), 0.3125, np.float32)
    out = sara_amd.apply_gaussian_filter(src, 2.0)
    assert np.allclose(out, 0.3125, rtol=0, atol=3e-7)


@pytest.mark.parametrize("shape", [(3, 3), (5, 64), (64, 5), (33, 65), (97, 131),
                                   (270, 480), (135, 240), (67, 119),
                                   # strip edges of the marching kernels (128- and
                                   # 256-column strips) and several row segments
                                   (40, 256), (90, 260), (50, 124), (300, 132),
                                   (26, 384),
                                   # widths that are not multiples of 4 on the
                                   # marching kernels (one full strip or more:
                                   # element-aligned rows, last strip moved left)
                                   (45, 683), (38, 1366), (70, 257), (52, 341),
                                   (41, 129), (33, 255)])
@pytest.mark.parametrize("sigma", [0.5, 1.2262735, 1.5198685, 1.946588, 2.4525296,
                                   3.0900156, 4.1])
def test_gaussian_filter_matches_oracle_bit_exact(oracle, shape, sigma):
    src = RNG.random(shape, dtype=np.float32)
    got = sara_amd.apply_gaussian_filter(src, sigma)
    want = oracle.apply_gaussian_filter(src, sigma)
    assert np.array_equal(got, want)


def test_gaussian_filter_generic_radius_path(oracle):
    # sigma = 5 -> 41 taps (radius 20 > 16): the runtime-radius kernel.
    src = RNG.random((150, 210), dtype=np.float32)
    assert len(sara_amd.make_gaussian_kernel(5.0)) == 41
    assert np.array_equal(sara_amd.apply_gaussian_filter(src, 5.0),
                          oracle.apply_gaussian_filter(src, 5.0))
    assert np.array_equal(sara_amd.apply_gaussian_filter(src, 7.9),
                          oracle.apply_gaussian_filter(src, 7.9))
    # up to 113 taps (radius 56: what 160 KB of LDS hold); sigma = 11.09 is the
    # largest increment of a 4-scale pyramid with k = 2 (89 taps)
    for sigma in (9.0, 11.085125, 14.0):
        assert np.array_equal(sara_amd.apply_gaussian_filter(src, sigma),
                              oracle.apply_gaussian_filter(src, sigma))
    with pytest.raises(sara_amd.SaraHipError):
        sara_amd.apply_gaussian_filter(src, 14.5)  # 117 taps > 113


def test_gaussian_filter_truncate_argument(oracle):
    src = RNG.random((40, 40), dtype=np.float32)
    for t in (1.0, 2.0, 1e-6):
        assert np.array_equal(sara_amd.apply_gaussian_filter(src, 1.6, t),
                              oracle.apply_gaussian_filter(src, 1.6, t))


# test_imageprocessing_resize.cpp:48-69
def test_downscale_table():
    src = np.array([[0, 0, 1, 1], [0, 0, 1, 1], [2, 2, 3, 3], [2, 2, 3, 3]],
                   np.float32)
    assert np.array_equal(sara_amd.downscale(src, 2), [[0, 1], [2, 3]])


@pytest.mark.parametrize("shape", [(135, 240), (67, 119), (9, 9), (270, 481)])
def test_downscale_matches_oracle(oracle, shape):
    src = RNG.random(shape, dtype=np.float32)
    assert np.array_equal(sara_amd.downscale(src, 2), oracle.downscale(src, 2))


# test_imageprocessing_resize.cpp:71-131
def test_enlarge_tables(oracle):
    src = np.array([[0, 1], [2, 3]], np.float32)
    true = np.array([[0, .5, 1, 1], [1, 1.5, 2, 2], [2, 2.5, 3, 3],
                     [2, 2.5, 3, 3]], np.float32)
    assert np.array_equal(sara_amd.enlarge(src, 4, 4), true)
    src = np.repeat(np.arange(5, dtype=np.float32)[:, None], 5, axis=1)
    true = np.repeat(np.array([0, .5, 1, 1.5, 2, 2.5, 3, 3.5, 4, 4],
                              np.float32)[:, None], 5, axis=1)
    assert np.linalg.norm(true - sara_amd.enlarge(src, 5, 10)) <= 1e-9
    with pytest.raises(sara_amd.SaraHipError) as e:
        sara_amd.enlarge(np.zeros((4, 4), np.float32), 2, 2)
    assert e.value.status == sara_amd.capi.OUT_OF_RANGE
    src = RNG.random((33, 47), dtype=np.float32)
    assert np.array_equal(sara_amd.enlarge(src, 94, 66), oracle.enlarge(src, 94, 66))
    assert np.array_equal(sara_amd.enlarge(src, 70, 50), oracle.enlarge(src, 70, 50))


# test_imageprocessing_differential.cpp:47-72 through the polar seam.
def test_polar_gradient_of_ramp():
    src = np.array([[1, 2, 3]] * 3, np.float32)
    g = sara_amd.gradient_polar_coordinates(src)
    for y in range(3):
        for x in range(3):
            assert g[y, x, 0] == 2 * (1.0 if x == 1 else 0.5)
            assert g[y, x, 1] == 0.0


@pytest.mark.parametrize("shape", [(2, 2), (3, 7), (64, 64), (135, 240), (101, 67),
                                   # odd widths and widths = 2 (mod 4) on the
                                   # marching kernel, tails in either pair group
                                   (40, 683), (35, 1366), (21, 257), (19, 385),
                                   (18, 129), (17, 127), (23, 255), (9, 5)])
def test_polar_gradient_matches_oracle_bit_exact(oracle, shape):
    src = RNG.random(shape, dtype=np.float32)
    got = sara_amd.gradient_polar_coordinates(src)
    want = oracle.gradient_polar(src)
    assert np.array_equal(got, want)  # glibc atan2f restated bit for bit


def test_polar_gradient_special_directions(oracle):
    # axis-aligned / diagonal / zero gradients exercise atan2f's branches.
    src = np.zeros((9, 9), np.float32)
    src[4, 4] = 1
    src[2, 6] = -3
    src[7, 1] = 2.5
    assert np.array_equal(sara_amd.gradient_polar_coordinates(src),
                          oracle.gradient_polar(src))


# test_imageprocessing_local_extremum.cpp:92-127 through the extremum map.
def test_extremum_map_predicates():
    I = np.ones((3, 10, 10), np.float32)
    m = sara_amd.scale_space_dog_extremum_map(I[0], I[1], I[2], 10.0, 0.01, 1)
    # plateau: non-strict max everywhere, but zero Hessian => on_edge rejects
    # (0 >= 0), so nothing survives.
    assert not m.any()
    I[1, 1, 1] = 10
    I[1, 7, 7] = 10
    m = sara_amd.scale_space_dog_extremum_map(I[0], I[1], I[2], 10.0, 0.01, 1)
    # the two spikes are the only maxima; their plateau neighbours qualify as
    # NON-strict minima (<= everything around them), as in the reference.
    assert m[1, 1] == 1 and m[7, 7] == 1 and np.count_nonzero(m == 1) == 2
    I[1, 1, 1] = -10
    I[1, 7, 7] = -10
    m = sara_amd.scale_space_dog_extremum_map(I[0], I[1], I[2], 10.0, 0.01, 1)
    assert m[1, 1] == -1 and m[7, 7] == -1 and np.count_nonzero(m == -1) == 2


@pytest.mark.parametrize("pad", [1, 3, 5])
def test_extremum_map_matches_oracle(oracle, pad):
    h, w = 60, 83
    layers = (RNG.random((3, h, w), dtype=np.float32) - 0.5) * 0.2
    # a few exact ties to exercise the non-strict comparisons
    layers[0, 20, 20:24] = layers[1, 20, 21]
    layers[2, 30, 40] = layers[1, 30, 40]
    got = sara_amd.scale_space_dog_extremum_map(layers[0], layers[1], layers[2],
                                                10.0, 0.01, pad)
    want = np.zeros((h, w), np.int8)
    for y in range(pad, h - pad):
        for x in range(pad, w - pad):
            t = oracle.scale_space_extremum(layers, x, y, strict=False)
            if t == 0:
                continue
            if abs(layers[1, y, x]) < np.float32(0.8) * np.float32(0.01):
                continue
            if oracle.on_edge(layers[1], x, y, 10.0):
                continue
            want[y, x] = t
    assert np.count_nonzero(want) > 20
    assert np.array_equal(got, want)


def test_subtract():
    lib = sara_amd.capi.load()
    import ctypes as C
    a = RNG.random(1000, dtype=np.float32)
    b = RNG.random(1000, dtype=np.float32)
    out = np.zeros(1000, np.float32)
    fp = C.POINTER(C.c_float)
    sara_amd.capi.check(lib.sara_hip_subtract(
        a.ctypes.data_as(fp), b.ctypes.data_as(fp), out.ctypes.data_as(fp),
        1000, 0))
    assert np.array_equal(out, a - b)


# ---- 8-bit frames converted on the device ------------------
def test_rgb8_to_gray32f_matches_reference_formula(oracle):
    rgb = RNG.integers(0, 256, size=(37, 53, 3), dtype=np.uint8)
    got = sara_amd.from_rgb8_to_gray32f(rgb)
    assert np.array_equal(got, oracle.rgb8_to_gray32f(rgb))
    # every (r, g, b) on a coarse lattice + the extremes
    v = np.array([0, 1, 2, 63, 64, 127, 128, 200, 254, 255], np.uint8)
    lat = np.stack(np.meshgrid(v, v, v, indexing="ij"), -1).reshape(50, 20, 3)
    assert np.array_equal(sara_amd.from_rgb8_to_gray32f(lat),
                          oracle.rgb8_to_gray32f(lat))
    g8 = RNG.integers(0, 256, size=(19, 31), dtype=np.uint8)
    assert np.array_equal(sara_amd.from_gray8_to_gray32f(g8),
                          g8.astype(np.float32) / np.float32(255))


def test_rgb8_sunflower_matches_golden_gray():
    import common
    from PIL import Image
    g = np.load(common.GOLDEN + "/sunflower_full.npz")
    rgb = np.array(Image.open(common.GOLDEN + "/sunflower_rgb8.png"))
    assert common.sha(sara_amd.from_rgb8_to_gray32f(rgb)) == str(g["gray_sha256"])


def test_short_device_math_is_exact_on_every_float():
    """The gradient kernel's short sqrt / division sequences and the look-up
    form of the atanf range reduction against the IEEE / select forms, on the
    device, for every non-negative float (sara_hip_selfcheck_device_math)."""
    import ctypes as C
    from sara_amd import capi
    bad = (C.c_ulonglong * 2)(1, 1)
    capi.check(capi.load().sara_hip_selfcheck_device_math(bad, 0))
    assert (bad[0], bad[1]) == (0, 0)


def test_orientation_bins_on_the_device_for_every_angle():
    """The orientation kernel's estimate-and-correct bin against the reference
    expression int(floor(double(ori / float(2 pi) * 36))) % 36, on the device,
    for all 1 086 918 620 floats of [0, float(2 pi)]
    (sara_hip_selfcheck_orientation_bins)."""
    import ctypes as C
    from sara_amd import capi
    bad = C.c_ulonglong(1)
    capi.check(capi.load().sara_hip_selfcheck_orientation_bins(C.byref(bad), 0))
    assert bad.value == 0


def _adversarial_hessians(rng, n):
    """Symmetric 3 x 3 float matrices built as Q diag(l) Q^T with spectra that
    are comfortably definite, indefinite, and - the interesting part - have
    their extreme eigenvalue within 1e-2 .. 1e-9 (relative) of zero on either
    side, at magnitudes 1e-8 .. 1e3; plus diagonal, rank-deficient and zero
    matrices."""
    q, _ = np.linalg.qr(rng.standard_normal((n, 3, 3)))
    lam = -np.abs(rng.standard_normal((n, 3))) - 0.05
    kind = rng.integers(0, 6, n)
    tiny = 10.0 ** rng.uniform(-9, -2, n) * rng.choice([-1.0, 1.0], n)
    lam[kind == 1, 0] = tiny[kind == 1]           # nearly singular
    lam[kind == 2] *= -1                          # positive definite
    lam[kind == 3, 1] *= -1                       # indefinite
    lam[kind == 4, 0] = 0.0                       # singular
    sel = kind == 5                               # two tiny eigenvalues
    lam[sel, 0] = tiny[sel]
    lam[sel, 1] = -tiny[sel] * rng.uniform(0.1, 10, sel.sum())
    mag = 10.0 ** rng.uniform(-8, 3, n)
    m = np.einsum("nij,nj,nkj->nik", q, lam, q) * mag[:, None, None]
    m = (m + m.transpose(0, 2, 1)) / 2
    m = m.astype(np.float32)
    m[:16] = 0
    for i in range(16, 48):                       # diagonal / axis-aligned
        m[i] = np.diag(np.diag(m[i]))
    return m


def test_definiteness_on_the_device_matches_the_oracle(oracle):
    """refine_extremum's definiteness decision as the extrema kernels take it
    (Sylvester shortcut in double, Eigen 3.4 float solver otherwise) against
    the oracle's restatement of the solver alone, on 600 000 matrices including
    nearly singular ones, for the three extremum types."""
    import ctypes as C
    from sara_amd import capi
    lib = capi.load()
    rng = np.random.default_rng(2026)
    n = 200_000
    for t in (1, 255, -1):
        m = _adversarial_hessians(rng, n)
        want = oracle.not_definite_enough3(m, t)
        types = np.full(n, t, np.int32)
        got = np.empty(n, np.uint8)
        capi.check(lib.sara_hip_selfcheck_definiteness(
            m.ctypes.data_as(C.POINTER(C.c_float)),
            types.ctypes.data_as(C.POINTER(C.c_int)), n,
            got.ctypes.data_as(C.POINTER(C.c_ubyte)), 0))
        assert np.array_equal(got.astype(bool), want), (
            t, int(np.count_nonzero(got.astype(bool) != want)))
        # both answers occur, so the comparison is not vacuous
        assert 0 < np.count_nonzero(want) < n


def test_extremum_map_halide_seam(oracle):
    """img_padding_sz = 0: the seam shakti_scale_space_dog_extremum_32f_cpu
    itself - every pixel, replicated borders, strict contrast, Halide hessian -
    against the oracle's restatement, exact."""
    h, w = 61, 87
    layers = (RNG.random((3, h, w), dtype=np.float32) - 0.5) * 0.2
    layers[1, 0, 5] = 0.8
    layers[1, h - 1, 0] = -0.8
    layers[0, 20, 20:24] = layers[1, 20, 21]
    layers[1, 40, 40] = np.float32(0.8) * np.float32(0.01)
    got = sara_amd.scale_space_dog_extremum_map(layers[0], layers[1], layers[2],
                                                10.0, 0.01, 0)
    want = oracle.halide_dog_extremum_map(layers[0], layers[1], layers[2], 10.0, 0.01)
    assert np.array_equal(got, want)
    assert got[0, 5] == 1 and got[h - 1, 0] == -1 and np.count_nonzero(got) > 20


def _tile_geometry(h, w, batch=1):
    """launch_blur_r (pyramid_kernels.hip): geometry from the number of 64 x 32
    tiles - 0: 64 x 32 / 512 threads, 1: 64 x 16 / 256, 2: 32 x 16 / 128."""
    tiles = ((w + 63) // 64) * ((h + 31) // 32) * batch
    return 0 if tiles >= 200 else (1 if tiles >= 48 else 2)


@pytest.mark.parametrize("geom", [0, 1, 2])
def test_tiled_blur_geometries_bit_exact(oracle, tmp_path, geom):
    """The tiled blur (what launches too small for the marching kernels use, and
    every launch of a one-frame call) has three tile geometries, picked from the
    number of tiles (64 x 32 / 512 threads, 64 x 16 / 256, 32 x 16 / 128).  Each
    is reached in a fresh process (SARA_HIP_BLUR=tile) through shapes of the
    matching tile count, with borders in every position of a tile, interior
    tiles (16-byte staging) and odd widths, for every radius of the pyramid and
    a few others; the fused half-size output is checked through a 3-octave
    pyramid."""
    import os
    import subprocess
    import sys
    shapes = {2: [(3, 3), (5, 64), (33, 65), (97, 131), (135, 240), (70, 517),
                  (41, 129), (50, 124), (66, 300), (17, 1366)],
              1: [(300, 517), (222, 1366), (270, 480), (129, 1027)],
              0: [(400, 1366), (517, 1027), (540, 960)]}[geom]
    assert all(_tile_geometry(h, w) == geom for h, w in shapes)
    sigmas = [0.5, 1.2262735, 1.5198685, 1.946588, 2.4525296, 3.0900156, 4.1, 0.9,
              2.2, 3.6]
    rng = np.random.default_rng(11 + geom)
    srcs = [rng.random(s, dtype=np.float32) for s in shapes]
    inp = tmp_path / "in.npz"
    out = tmp_path / "out.npz"
    np.savez(inp, **{"s%d" % i: a for i, a in enumerate(srcs)})
    script = tmp_path / "run.py"
    script.write_text(
        "import sys, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "import sara_amd\n"
        "from sara_amd.synth import synth\n"
        "d = np.load(%r)\n"
        "sig = %r\n"
        "res = {}\n"
        "for i in range(len(d.files)):\n"
        "    for j, s in enumerate(sig):\n"
        "        res['b%%d_%%d' %% (i, j)] = sara_amd.apply_gaussian_filter(d['s%%d' %% i], s)\n"
        "img = synth(301, 222, 99)\n"
        "p = sara_amd.ImagePyramidParams(0, 6, num_octaves_max=3)\n"
        "with sara_amd.SiftContext(301, 222, 1, p) as ctx:\n"
        "    ctx.detect(img)\n"
        "    for o in range(3):\n"
        "        for s in range(6):\n"
        "            res['g%%d_%%d' %% (o, s)] = ctx.gaussian(s, o, 0)\n"
        "np.savez(%r, **res)\n"
        % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), str(inp),
           sigmas, str(out)))
    env = dict(os.environ)
    env["SARA_HIP_BLUR"] = "tile"
    subprocess.run([sys.executable, str(script)], check=True, env=env)
    got = np.load(out)
    for i, src in enumerate(srcs):
        for j, s in enumerate(sigmas):
            assert np.array_equal(got["b%d_%d" % (i, j)],
                                  oracle.apply_gaussian_filter(src, s)), (shapes[i], s)
    from sara_amd.synth import synth
    ref = oracle.RefSift(synth(301, 222, 99),
                         oracle.PyramidParams(0, 6, None, 1, 0.5, 1.6, 3))
    for o in range(3):
        for s in range(6):
            assert np.array_equal(got["g%d_%d" % (o, s)], ref.gaussian(s, o)), (o, s)
