"""The real-image parity pack (tests/golden/real/*.png, real_images.npz; made
by tests/golden/make_real_images.py from the photographs under the
reference's data/) pins the oracle against drift on real inputs - saturated
regions, JPEG blocks, long straight edges, plateaus - and the oracle's
exhaustive matcher against what the reference's vendored FLANN (linear index)
returned for the examples' matching pair
(examples/Sara/FeatureMatching/image_sift_matching.cpp:25-60)."""
import numpy as np
import pytest

import real_images as ri


@pytest.mark.parametrize("name", ri.NAMES)
@pytest.mark.parametrize("tag", ri.TAGS)
def test_oracle_reproduces_the_pack(oracle, name, tag):
    g = ri.pack()
    gray = ri.gray(oracle, name)
    r = oracle.RefSift(gray, ri.ref_params(oracle, tag), parallel=True)
    reg, so, desc = r.keypoints()
    ereg, exyso = r.extrema()
    k = "%s_%s_" % (name, tag)
    assert r.octave_count == int(g[k + "octaves"])
    assert np.array_equal(reg.view(np.uint8).reshape(-1, 48), g[k + "regions"])
    assert np.array_equal(so, g[k + "scale_octave"])
    assert np.array_equal(exyso, g[k + "extrema_xyso_type"])
    assert np.array_equal(ereg.view(np.uint8).reshape(-1, 48),
                          g[k + "extrema_regions"])
    assert np.array_equal(desc[::8], g[k + "desc_every8"])
    assert np.array_equal(desc.sum(axis=1), g[k + "desc_row_sums"])


@pytest.mark.parametrize("ratio", ri.RATIOS)
def test_oracle_matcher_equals_flann_on_the_examples_pair(oracle, ratio):
    g = ri.pack()
    d1, d2 = ri.pair_descriptors(oracle)
    got = oracle.compute_matches(d1, d2, ratio)
    want = g["pair_matches_%.1f" % ratio]
    assert len(got) == len(want)
    assert got.tobytes() == want.tobytes()
