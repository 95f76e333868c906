"""The reference's own hot-path unit tests, restated
against the CPU oracle.  Each test cites the reference test it restates
(paths relative to /root/reference/cpp/test/Sara).  These are the only result
pins the reference offers; they run on CPU.
"""
import math

import numpy as np
import pytest

import refbind as rb


# ImageProcessing/test_imageprocessing_linear_filtering.cpp:28-43
def test_convolve_array(oracle):
    out = oracle.convolve_array(np.ones(10), np.ones(3), 8)
    assert np.array_equal(out, [3] * 8 + [1] * 2)


SRC3 = np.array([[1, 2, 3]] * 3, dtype=np.float32)
KDIFF = np.array([-0.5, 0.0, 0.5], dtype=np.float32)


# ...linear_filtering.cpp:69-100
def test_row_based_filter(oracle):
    out = oracle.apply_row_based_filter(SRC3, KDIFF)
    assert np.array_equal(out, np.array([[0.5, 1, 0.5]] * 3, dtype=np.float32))


def test_column_based_filter(oracle):
    out = oracle.apply_column_based_filter(SRC3, KDIFF)
    assert np.array_equal(out, np.zeros((3, 3), dtype=np.float32))


def _true_gaussian(n):
    c = n // 2
    i, j = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    m = np.exp(-((i - c) ** 2 + (j - c) ** 2).astype(np.float32) / np.float32(2))
    return (m / m.sum()).astype(np.float32)


# ...linear_filtering.cpp:136-187: Dirac responses, L2 distance 1e-5.
@pytest.mark.parametrize("n,truncate", [(3, 1.0), (9, 4.0), (65, 4.0)])
def test_gaussian_on_dirac(oracle, n, truncate):
    src = np.zeros((n, n), dtype=np.float32)
    src[n // 2, n // 2] = 1
    out = oracle.apply_gaussian_filter(src, 1.0, truncate)
    if n == 65:
        # kernel is 9 taps; the rest of the 65x65 true matrix is < 1e-7.
        true = np.zeros((65, 65), dtype=np.float64)
        i, j = np.meshgrid(np.arange(65), np.arange(65), indexing="ij")
        true = np.exp(-((i - 32.0) ** 2 + (j - 32.0) ** 2) / 2.0)
        true /= true.sum()
    else:
        true = _true_gaussian(n)
    assert np.linalg.norm(true - out) < 1e-5


# kernel size rule of LinearFiltering.hpp:171-203.
def test_gaussian_kernel_sizes(oracle):
    k = np.float32(2.0) ** np.float32(1.0 / 3.0)
    s = np.float32(1.6)
    sizes = []
    for _ in range(5):
        sigma = np.float32(math.sqrt(float(np.float32(k * s) ** 2 - s * s)))
        sizes.append(len(oracle.make_gaussian_kernel(float(sigma))))
        s = np.float32(s * k)
    assert sizes == [11, 13, 17, 21, 25]
    init = np.float32(math.sqrt(1.6 ** 2 - 0.5 ** 2))
    assert len(oracle.make_gaussian_kernel(float(init))) == 13
    init1 = np.float32(np.sqrt(np.float32(1.6) ** 2 - np.float32(1.0)))
    assert len(oracle.make_gaussian_kernel(float(init1))) == 11
    for sig in (0.1, 1.0, 2.54):
        kern = oracle.make_gaussian_kernel(sig)
        assert len(kern) % 2 == 1 and len(kern) >= 3
        assert abs(kern.sum() - 1) < 1e-6


# ImageProcessing/test_imageprocessing_resize.cpp:48-69
def test_downscale(oracle):
    src = np.array([[0, 0, 1, 1], [0, 0, 1, 1], [2, 2, 3, 3], [2, 2, 3, 3]],
                   dtype=np.float32)
    assert np.array_equal(oracle.downscale(src, 2), [[0, 1], [2, 3]])


# ...resize.cpp:71-131
def test_enlarge(oracle):
    src = np.array([[0, 1], [2, 3]], dtype=np.float32)
    true = np.array([[0, 0.5, 1, 1], [1, 1.5, 2, 2], [2, 2.5, 3, 3],
                     [2, 2.5, 3, 3]], dtype=np.float32)
    assert np.array_equal(oracle.enlarge(src, 4, 4), true)
    src = np.repeat(np.arange(5, dtype=np.float32)[:, None], 5, axis=1)
    out = oracle.enlarge(src, 5, 10)
    true = np.repeat(np.array([0, .5, 1, 1.5, 2, 2.5, 3, 3.5, 4, 4],
                              dtype=np.float32)[:, None], 5, axis=1)
    assert np.linalg.norm(true - out) <= 1e-9
    with pytest.raises(ValueError):
        oracle.enlarge(np.zeros((4, 4)), 2, 2)


# ImageProcessing/test_imageprocessing_differential.cpp:47-72
def test_gradient(oracle):
    g = oracle.gradient(SRC3)
    for y in range(3):
        for x in range(3):
            assert g[y, x, 0] == (1 if x == 1 else 0.5)
            assert g[y, x, 1] == 0


# ...differential.cpp:95-122
def test_hessian_of_constant(oracle):
    assert np.array_equal(oracle.hessian(np.ones((3, 3))), np.zeros((3, 3, 3)))


# ImageProcessing/test_imageprocessing_local_extremum.cpp:92-127
def test_local_scale_space_extremum(oracle):
    I = np.ones((3, 10, 10), dtype=np.float32)
    assert oracle.scale_space_extremum(I, 1, 1, strict=True) == 0
    # non-strict on a plateau: max test fires first.
    assert oracle.scale_space_extremum(I, 1, 1, strict=False) == 1
    I[1, 1, 1] = 10
    I[1, 7, 7] = 10
    assert oracle.scale_space_extremum(I, 1, 1, strict=True) == 1
    assert oracle.scale_space_extremum(I, 7, 7, strict=True) == 1
    n_max = sum(oracle.scale_space_extremum(I, x, y, strict=True) == 1
                for y in range(1, 9) for x in range(1, 9))
    assert n_max == 2
    I[1, 1, 1] *= -1
    I[1, 7, 7] *= -1
    assert oracle.scale_space_extremum(I, 1, 1, strict=False) == -1
    assert oracle.scale_space_extremum(I, 1, 1, strict=True) == -1
    n_min = sum(oracle.scale_space_extremum(I, x, y, strict=True) == -1
                for y in range(1, 9) for x in range(1, 9))
    assert n_min == 2


# ImageProcessing/test_imageprocessing_gaussian_pyramid.cpp:30-49
def test_gaussian_pyramid_with_fixed_octaves(oracle):
    I = np.ones((16, 16), dtype=np.float32)
    p = oracle.PyramidParams(-1, 2, 2.0, 1, 0.5, 1.6, 2)
    r = oracle.RefSift(I, p, pyramid_only=True)
    assert r.octave_count == 2
    assert r.octave_info(0)[:2] == (32, 32)
    assert r.octave_info(1)[:2] == (16, 16)


# ...gaussian_pyramid.cpp:51-58 (builds for ImagePyramidParams(-1)).
def test_gaussian_pyramid_default_params(oracle):
    I = np.ones((16, 16), dtype=np.float32)
    r = oracle.RefSift(I, oracle.PyramidParams(-1), pyramid_only=True)
    # l = 32, b = 1 -> int(log(16)/log(2)) = 4 octaves: 32, 16, 8, 4.
    assert r.octave_count == 4
    assert [r.octave_info(o)[0] for o in range(4)] == [32, 16, 8, 4]
    for o in range(4):
        for s in range(5):
            assert np.all(np.abs(r.dog(s, o)) < 1e-6)


# FeatureDetectors/test_featuredetectors_dog.cpp:45-100
def test_compute_dog_extrema_blob(oracle):
    N = 11
    I = np.zeros((N, N), dtype=np.float32)
    I[3:8, 3:8] = 1
    k = float(np.power(np.float32(2.0), np.float32(1.0) / np.float32(3)))
    p = oracle.PyramidParams(0, 6, k, 1, 1.0, 1.6)
    # ComputeDoGExtrema{pyramid_params, 1e-6f, 1e-6f}: truncate, threshold;
    # edge ratio 10, padding 1, 5 iterations by default.  The driver
    # compute_sift_keypoints shifts extremum_refinement_iter into the padding
    # slot (Q1), so padding 1 is requested with extremum_refinement_iter=1.
    r = oracle.RefSift(I, p, gauss_truncate=1e-6, extremum_thres=1e-6,
                       edge_ratio_thres=10.0, extremum_refinement_iter=1,
                       stop_after=3)
    regions, xyso = r.extrema()
    assert len(regions) > 0
    f = regions[0]
    z = r.octave_info(int(xyso[0, 3]))[2]
    assert abs(f["coords"][0] * z - 5) < 1e-2
    assert abs(f["coords"][1] * z - 5) < 1e-2


# FeatureDescriptors/test_featuredescriptors_orientation.cpp:26-50
def test_lowe_smooth_histogram(oracle):
    h = np.zeros(36, dtype=np.float32)
    h[0] = 1
    h[14] = 1
    h = oracle.lowe_smooth_histogram(h, 1)
    for i in (35, 0, 1, 13, 14, 15):
        assert abs(h[i] - 1 / 3) < 1e-5 / 3


# ...orientation.cpp:52-99: hard binning into M = 24 bins.
def test_orientation_histogram(oracle):
    N, M = 5, 24
    c = np.float32(2.5)
    for gy in range(N):
        for gx in range(N):
            t = np.float32(math.atan2(float(np.float32(gy) - c),
                                      float(np.float32(gx) - c)))
            if t < 0:
                t = np.float32(t + np.float32(2) * np.float32(math.pi))
            theta_bin = int(math.floor(t / np.float32(2 * math.pi) * M)) % M
            grad = np.zeros((N, N, 2), dtype=np.float32)
            grad[gy, gx] = (1.0, t)
            hist = oracle.orientation_histogram(grad, c, c, 1.0, bins=M)
            hist = hist / hist.sum()
            expected = np.zeros(M, dtype=np.float32)
            expected[theta_bin] = 1
            assert np.linalg.norm(expected - hist) < 1e-6


# ...orientation.cpp:101-124
def test_detect_single_peak(oracle):
    N = 5
    c = np.float32(2.5)
    theta = np.float32(math.atan2(0 - 2.5, 0 - 2.5))
    grad = np.zeros((N, N, 2), dtype=np.float32)
    grad[0, 0] = (1.0, theta)
    peaks, _ = oracle.dominant_orientations(grad, c, c, 1.0)
    assert len(peaks) == 1
    assert abs(theta - peaks[0]) < 1e-6


# FeatureDescriptors/test_featuredescriptors_sift.cpp:25-55
def test_sift_descriptor_nonzero(oracle):
    N = 5
    c = np.float32(2.5)
    theta = np.float32(math.atan2(0 - 2.5, 0 - 2.5))
    grad = np.zeros((N, N, 2), dtype=np.float32)
    grad[0, 0] = (1.0, theta)
    # OERegion{c, 1.f}.scale() == 1; orientation 0.
    assert oracle.lib().ref_oeregion_scale(1.0) == 1.0
    d = oracle.sift_descriptor(grad, c, c, 1.0, 0.0)
    assert d.shape == (128,)
    assert np.any(d != 0)
    assert np.all(d <= 255) and np.all(d >= 0)


# Core/Pixel colour conversion used for config 1 (SmartColorConversion.hpp:237-246).
def test_rgb_to_gray(oracle):
    rng = np.random.default_rng(0)
    rgb = rng.integers(0, 256, size=(64, 3), dtype=np.uint8)
    vec = oracle.rgb8_to_gray32f(rgb)
    for i in range(64):
        r, g, b = (int(v) for v in rgb[i])
        assert vec[i] == np.float32(oracle.lib().ref_rgb8_to_gray32f(r, g, b))
    assert oracle.rgb8_to_gray32f(np.array([255, 255, 255])) == np.float32(1.0)


# Features/Feature.hpp:79-83 + Feature.cpp:28-39: scale() inverts the
# constructor within float rounding.
def test_oeregion_scale_roundtrip(oracle):
    for s in (1.6, 2.0159, 3.2, 5.0797, 1.7342):
        got = oracle.lib().ref_oeregion_scale(s)
        assert abs(got - np.float32(s)) <= 2e-7 * s


# ---- descriptor matching ----------------------
def test_ann_matching_reference_case(oracle):
    """test_featurematching_matching.cpp:29-62: one point (0, 0) against ten
    points (i, i), ratio 0.6 -> exactly one match {0, 0} with score 0."""
    d1 = np.zeros((1, 2), np.float32)
    d2 = np.stack([np.arange(10), np.arange(10)], axis=1).astype(np.float32)
    m = oracle.compute_matches(d1, d2, 0.6)
    assert len(m) == 1
    assert (m[0]["x_index"], m[0]["y_index"], m[0]["score"]) == (0, 0, 0.0)


def test_matching_restatement_properties(oracle):
    rng = np.random.default_rng(11)
    d1 = rng.random((40, 128), dtype=np.float32)
    d2 = np.concatenate([d1[:25] + rng.normal(0, 1e-3, (25, 128)).astype(np.float32),
                         rng.random((30, 128), dtype=np.float32)])
    m = oracle.compute_matches(d1, d2, 0.6)
    # the 25 perturbed copies are mutual nearest neighbours, found from both
    # sides and kept once (AnnMatcher.cpp:239-254)
    pairs = {(int(a), int(b)) for a, b in zip(m["x_index"], m["y_index"])}
    assert pairs == {(i, i) for i in range(25)}
    assert np.all(np.diff(m["score"]) >= 0)
    assert np.all(m["score"] <= np.float32(0.6) * np.float32(0.6))
    # FLANN's distance: groups of four, float accumulator (dist.h:150-178)
    a, b = d1[3], d2[7]
    acc = np.float32(0)
    for i in range(0, 128, 4):
        d = a[i:i + 4] - b[i:i + 4]
        acc = acc + (((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]) + d[3] * d[3])
    assert oracle.flann_l2(a, b) == float(acc)
    with pytest.raises(RuntimeError):
        oracle.compute_matches(np.zeros((0, 128), np.float32), d2, 0.6)


def test_matching_default_ratio_radius_search(oracle):
    """AnnMatcher.cpp:133-154 with the default sift_ratio_thres = 1.2f, worked
    by hand.  Query q0 = (0); candidates at squared distances 4, 5, 5.7, 6, 9:
    radius = 4 * 1.44 = 5.76 -> K = 3 (strictly inside); rank 1 scores
    d0 / d1 = 0.8, ranks 2, 3 score d / d0 = 1.25, 1.425.  A second query q1 =
    (2.3) sits next to the candidates, so that in the other direction each of
    them finds q1, not q0, and the (0, j) pairs come from q0's search alone."""
    d1 = np.array([[0.0], [2.3]], np.float32)
    d2 = np.sqrt(np.array([[4.0], [5.0], [5.7], [6.0], [9.0]])).astype(np.float32)
    m = oracle.compute_matches(d1, d2, 1.2)
    q0 = m[m["x_index"] == 0]
    q0 = q0[np.argsort(q0["rank"])]
    assert list(q0["y_index"]) == [0, 1, 2] and list(q0["rank"]) == [1, 2, 3]
    assert list(q0["direction"]) == [0, 0, 0]
    dd = (d2[:, 0] * d2[:, 0]).astype(np.float32)
    assert q0["score"][0] == dd[0] / dd[1]
    assert q0["score"][1] == dd[1] / dd[0] and q0["score"][2] == dd[2] / dd[0]
    # q1's best neighbour is candidate 1 (found from both sides, kept once with
    # the lower score, :239-254); every candidate's best neighbour is q1
    q1 = m[m["x_index"] == 1]
    assert sorted(q1["y_index"]) == [0, 1, 2, 3, 4] and np.all(q1["rank"] == 1)
    assert np.all(np.diff(m["score"]) >= 0)
    assert np.all(m["score"] <= np.float32(1.2) * np.float32(1.2))
    # one candidate only: score 1 (:87-101), kept because 1 < 1.44
    one = oracle.compute_matches(d2, d1[:1], 1.2)
    assert len(one) == 5 and np.all(one["y_index"] == 0)
    # ... except for the pair the lone key's own search also finds (score 0.8)
    assert sorted(one["score"]) == [np.float32(4) / np.float32(5), 1, 1, 1, 1]
    # a best distance of exactly 0 gives radius 0: K = 0, nothing from that side
    z = oracle.compute_matches(np.zeros((2, 2), np.float32),
                               np.stack([np.arange(10.0), np.arange(10.0)], 1)
                               .astype(np.float32), 1.2)
    assert not (z["direction"] == 0).any()


def test_self_matching_restatement(oracle):
    """AnnMatcher{keys, ...} (AnnMatcher.cpp:199-215): rank 0 is the key itself,
    ranks >= 1 are emitted inside the radius unless KeyProximity
    (KeyProximity.cpp:17-30) finds the two keys too close; nothing at all for
    ratio <= 1 (the loop runs over [1, K = 1)).  Reference pin:
    test_featurematching_key_proximity.cpp:26-37."""
    f1 = np.array([0, 0, 1, 0, 0, 1, 0, 11], np.float32)
    f2 = np.array([0, 0.1, 1 / 1.1 ** 2, 0, 0, 1 / 1.1 ** 2, 0, 11], np.float32)
    assert oracle.key_proximity(f1, f2)
    far = np.array([100, 0, 1, 0, 0, 1, 0, 11], np.float32)
    assert not oracle.key_proximity(f1, far)
    assert oracle.key_proximity(f1, far, 0.5, 101.0)        # pixel threshold
    wide = np.array([100, 0, 1e-6, 0, 0, 1e-6, 0, 11], np.float32)
    assert oracle.key_proximity(f1, wide)                   # its metric: 0.01 < 0.25
    # four keys on a line, descriptors = positions: 0, 1, 2.1, 50 (squared
    # distances from key 0: 1, 4.41, 2500), 40 px apart in the image
    d = np.array([[0.0], [1.0], [2.1], [50.0]], np.float32)
    reg = np.zeros(4, [("coords", "<f4", 2), ("shape_matrix", "<f4", 4),
                       ("orientation", "<f4"), ("type", "u1")])
    reg["coords"][:, 0] = [0, 40, 80, 120]
    reg["shape_matrix"] = [1, 0, 0, 1]
    m = oracle.compute_self_matches(d, reg, 3.0)            # ratio^2 = 9
    got = {(int(a), int(b)): (float(s), int(r)) for a, b, s, r in
           zip(m["x_index"], m["y_index"], m["score"], m["rank"])}
    # key 0: neighbours 1 (d 1) and 2 (d 4.41 < 9): rank 1 scores 1 / 4.41
    dd = np.float32(2.1) * np.float32(2.1)
    assert got[(0, 1)][1] == 1 and got[(0, 1)][0] == float(np.float32(1) / dd)
    assert (0, 2) in got and got[(0, 2)][1] == 2
    assert all(a != b for a, b in got)                      # never a key with itself
    assert len(oracle.compute_self_matches(d, reg, 1.0)) == 0
    # with a 50 px pixel threshold the adjacent keys are redundant
    m2 = oracle.compute_self_matches(d, reg, 3.0, 0.5, 50.0)
    assert all(abs(int(a) - int(b)) > 1 for a, b in zip(m2["x_index"], m2["y_index"]))


def test_root_sift_restatement(oracle):
    """FeatureDescriptors/RootSIFT.hpp:45-53 (dead code in the reference: Eigen 2
    API) - h /= lpNorm<1>(h); h = sqrt(h).  Known answers by hand."""
    d = np.array([[1, 3, 0, 12], [0, 0, 0, 0], [2, -2, 2, 2]], np.float32)
    got = oracle.root_sift(d)
    want = np.array([[0.25, np.sqrt(np.float32(3) / np.float32(16)), 0,
                      np.sqrt(np.float32(0.75))], [0, 0, 0, 0],
                     [0.5, -0.5, 0.5, 0.5]], np.float32)
    assert np.array_equal(got, want)
    # Hellinger kernel: <root(a), root(b)> = sum sqrt(a_i b_i) / sqrt(|a|_1 |b|_1)
    rng = np.random.default_rng(0)
    a, b = rng.random((2, 128), dtype=np.float32)
    ra, rb = oracle.root_sift(a[None])[0], oracle.root_sift(b[None])[0]
    assert abs(float(ra @ rb) - np.sqrt(a * b).sum() / np.sqrt(a.sum() * b.sum())) < 1e-5


def test_corrected_mode_switches_of_the_restatement(oracle):
    """SURVEY.md section 8f row f4.  Default: quirk Q2 (minima never refined:
    integer positions) and Q3 (octave o+1 from G(2, o)).  Signed extremum type
    (host loop of RefineExtremum.cpp:226-361): minima refined; downscale at the
    doubled sigma: octave 1 is the nearest-neighbour half of G(3, 0)."""
    from sara_amd.synth import synth
    img = synth(200, 160, 3)
    p = oracle.PyramidParams(0, 6, None, 1, 0.5, 1.6, 3)
    base = oracle.RefSift(img, p, stop_after=3)
    reg = base.extrema()[0]
    mins = reg[reg["extremum_type"] == -1]["coords"]
    assert len(mins) > 10 and np.all(mins == np.round(mins))
    assert np.array_equal(base.gaussian(0, 1), base.gaussian(2, 0)[::2, ::2][:80, :100])
    with oracle.detector_mode(oracle.MODE_SIGNED_EXTREMUM_TYPE):
        signed = oracle.RefSift(img, p, stop_after=3)
    sreg = signed.extrema()[0]
    smins = sreg[sreg["extremum_type"] == -1]["coords"]
    assert np.any(smins != np.round(smins))
    # Maxima: the branch classifies with the Halide rules (every pixel, strict
    # contrast, its own hessian in the edge test - more lenient on this image),
    # refines like the default build and drops implausible refined scales
    # (:307-325): a default maximum that survives has the same record.
    smax, dmax = sreg[sreg["extremum_type"] == 1], reg[reg["extremum_type"] == 1]
    kept = {(c.tobytes(), m.tobytes(), float(v)) for c, m, v in
            zip(smax["coords"], smax["shape_matrix"], smax["extremum_value"])}
    common_max = [(c.tobytes(), m.tobytes(), float(v)) in kept for c, m, v in
                  zip(dmax["coords"], dmax["shape_matrix"], dmax["extremum_value"])]
    assert len(smax) > 0 and sum(common_max) >= 0.8 * len(dmax)
    assert np.array_equal(signed.gaussian(0, 1), base.gaussian(0, 1))
    with oracle.detector_mode(oracle.MODE_DOWNSCALE_AT_DOUBLE_SIGMA):
        fixed = oracle.RefSift(img, p, stop_after=3)
    assert np.array_equal(fixed.gaussian(0, 1), fixed.gaussian(3, 0)[::2, ::2][:80, :100])
    assert np.array_equal(fixed.gaussian(3, 0), base.gaussian(3, 0))
    # and the mode is restored
    again = oracle.RefSift(img, p, stop_after=3)
    assert again.extrema()[0].tobytes() == reg.tobytes()
