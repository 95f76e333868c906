"""-m gpu: the HIP path on real photographs - the images the reference's own
examples and tests read (tests/golden/real/, from /root/reference/data) -
against the committed oracle outputs (tests/golden/real_images.npz,
tests/golden/make_real_images.py).  Saturated regions, JPEG blocks, long
straight edges and real plateaus (the non-strict `>=` of the extremum test,
SURVEY Q10) are where a GPU scan and a CPU scan would drift apart.

Bars as in test_gpu_pipeline.py: extremum sites, order, (s, o), coordinates,
values exact; shape matrix rel 1e-6; orientation 1e-6 rad; descriptors max-abs
2e-3 (of 0..255).  The matcher is compared BYTE FOR BYTE with what the
reference's vendored FLANN (exact index) returned for the examples' pair
(examples/Sara/FeatureMatching/image_sift_matching.cpp:25-60)."""
import numpy as np
import pytest

import common
import real_images as ri
import sara_amd
from test_gpu_pipeline import DESC_ATOL, SHAPE_RTOL, THETA_ATOL

pytestmark = pytest.mark.gpu


def assert_keys_equal_pack(keys, name, tag):
    g = ri.pack()
    k = "%s_%s_" % (name, tag)
    want = common.regions_from_bytes(g[k + "regions"])
    assert len(keys) == len(want) > 0
    assert np.array_equal(keys.scale_octave, g[k + "scale_octave"])
    common.assert_regions_equal(keys.regions, want, rtol_shape=SHAPE_RTOL,
                                atol_theta=THETA_ATOL)
    d = keys.descriptor_matrix
    worst = float(np.max(np.abs(d[::8] - g[k + "desc_every8"])))
    assert worst <= DESC_ATOL
    assert np.max(np.abs(d.sum(axis=1) - g[k + "desc_row_sums"])) <= 128 * DESC_ATOL
    return worst


@pytest.mark.parametrize("name", ri.NAMES)
@pytest.mark.parametrize("tag", ri.TAGS)
def test_float_frames_through_the_free_function(oracle, name, tag, kernel_selection):
    """image.convert<float>() -> compute_sift_keypoints(), as the example."""
    gray = ri.gray(oracle, name)
    keys = sara_amd.compute_sift_keypoints(gray, ri.hip_params(tag))
    worst = assert_keys_equal_pack(keys, name, tag)
    print("%s/%s: %d keypoints, descriptors within %.1e" % (name, tag, len(keys), worst))


@pytest.mark.parametrize("name", ri.NAMES)
@pytest.mark.parametrize("tag", ri.TAGS)
def test_extrema_through_compute_dog_extrema(name, tag, kernel_selection):
    import refbind as rb
    g = ri.pack()
    k = "%s_%s_" % (name, tag)
    gray = ri.gray(rb, name)
    # compute_sift_keypoints hands extremum_refinement_iter = 5 over as the
    # border padding (SIFT.cpp:45-51, Q1)
    dog = sara_amd.ComputeDoGExtrema(ri.hip_params(tag), img_padding_sz=5)
    regions, so = dog(gray)
    want = common.regions_from_bytes(g[k + "extrema_regions"])
    assert len(regions) == len(want) > 0
    common.assert_regions_equal(regions, want, rtol_shape=SHAPE_RTOL)
    assert np.array_equal(so, g[k + "extrema_xyso_type"][:, 2:4])


@pytest.mark.parametrize("name", ri.NAMES)
def test_rgb8_frames_converted_on_the_device(oracle, name, kernel_selection):
    """The f1 upload path: interleaved RGB8 in, converted by the device with
    the reference's arithmetic - byte-identical to the float path."""
    rgb = ri.rgb(name)
    gray = ri.gray(oracle, name)
    h, w = gray.shape
    for tag in ri.TAGS:
        with sara_amd.SiftContext(w, h, 1, ri.hip_params(tag)) as ctx:
            a = ctx.detect_u8(rgb).keypoint_lists()[0]
            b = ctx.detect(gray).keypoint_lists()[0]
        assert a.regions.tobytes() == b.regions.tobytes()
        assert a.descriptor_matrix.tobytes() == b.descriptor_matrix.tobytes()
        assert_keys_equal_pack(a, name, tag)


@pytest.mark.parametrize("ratio", ri.RATIOS)
def test_matcher_on_the_examples_pair_equals_flann(oracle, ratio):
    """AnnMatcher{keys1, keys2, ratio}.compute_matches() on the oracle's
    descriptors of All / GuardOnBlonde: the bytes FLANN's linear index gave
    (11 / 3 697 / 139 724 matches - the last one is the adaptive radius search
    of an unrelated pair, 50 members per key)."""
    d1, d2 = ri.pair_descriptors(oracle)
    want = ri.pack()["pair_matches_%.1f" % ratio]
    got = sara_amd.AnnMatcher(d1, d2, ratio).compute_matches()
    assert len(got) == len(want)
    assert got.tobytes() == want.tobytes()


@pytest.mark.parametrize("ratio", ri.RATIOS)
def test_front_end_pair_on_the_examples_pair(oracle, ratio):
    """Detection AND matching on the device (both frames in one batch, the
    descriptors matched where they are in HBM).  The GPU's descriptors are
    within 2e-3 of the oracle's, not bit-equal, so a match whose score sits on
    the threshold may come or go: the lists agree on all but a handful of
    pairs and on the scores of the common ones."""
    want = ri.pack()["pair_matches_%.1f" % ratio]
    frames = np.stack([ri.gray(oracle, n) for n in ri.PAIR])
    _, h, w = frames.shape
    with sara_amd.SiftContext(w, h, 2, ri.hip_params("default")) as ctx:
        ctx.detect(frames)
        counts, _ = ctx.counts()
        got = ctx.match_frames(0, 1, ratio)
    g = ri.pack()
    assert [int(c) for c in counts] == [len(g["All_default_regions"]),
                                        len(g["GuardOnBlonde_default_regions"])]

    def keyed(m):
        return {(int(a), int(b)): float(s)
                for a, b, s in zip(m["x_index"], m["y_index"], m["score"])}
    kg, kw = keyed(got), keyed(want)
    common_pairs = set(kg) & set(kw)
    only = len(set(kg) ^ set(kw))
    print("ratio %.1f: %d matches on the device, %d from FLANN on the oracle's "
          "descriptors, %d not in both" % (ratio, len(kg), len(kw), only))
    assert only <= max(2, len(kw) // 500)
    sg = np.array([kg[p] for p in common_pairs])
    sw = np.array([kw[p] for p in common_pairs])
    assert np.allclose(sg, sw, rtol=2e-4, atol=1e-7)
