"""-m gpu: descriptor matching through the C-ABI
against the oracle's exhaustive-search restatement of AnnMatcher (both
constructors; ratios below and above 1, i.e. with and without the adaptive
radius search of AnnMatcher.cpp:133-138).  Bar: identical match lists - indices,
ranks, directions - and bit-identical scores (FLANN's squared L2 summation
order is reproduced on the device)."""
import numpy as np
import pytest

import sara_amd
from sara_amd.synth import synth

pytestmark = pytest.mark.gpu


def assert_same(got, want):
    assert len(got) == len(want)
    assert np.array_equal(got["x_index"], want["x_index"])
    assert np.array_equal(got["y_index"], want["y_index"])
    assert np.array_equal(got["score"], want["score"])
    assert np.array_equal(got["rank"], want["rank"])
    assert np.array_equal(got["direction"], want["direction"])


def test_reference_matcher_case():
    """test_featurematching_matching.cpp:29-62."""
    d1 = np.zeros((1, 2), np.float32)
    d2 = np.stack([np.arange(10), np.arange(10)], axis=1).astype(np.float32)
    m = sara_amd.AnnMatcher(d1, d2, 0.6).compute_matches()
    assert len(m) == 1
    assert (m[0]["x_index"], m[0]["y_index"], m[0]["score"]) == (0, 0, 0.0)


@pytest.mark.parametrize("n1,n2,dim", [(40, 55, 128), (300, 129, 128), (65, 64, 32),
                                       (7, 200, 6), (2, 2, 128), (1, 70, 128),
                                       (130, 1, 128), (33, 500, 127)])
def test_random_descriptors_match_oracle(oracle, n1, n2, dim):
    rng = np.random.default_rng(n1 * 1000 + n2)
    d1 = rng.random((n1, dim), dtype=np.float32)
    k = min(n1, n2) // 2
    d2 = rng.random((n2, dim), dtype=np.float32)
    d2[:k] = d1[:k] + rng.normal(0, 2e-3, (k, dim)).astype(np.float32)
    # 1.2 is the reference's DEFAULT (AnnMatcher.hpp:36-46): the adaptive radius
    # search of AnnMatcher.cpp:133-138, matches of rank > 1
    for ratio in (0.6, 0.9, 1.0, 1.05, 1.2, 2.0):
        got = sara_amd.match(d1, d2, ratio)
        assert_same(got, oracle.compute_matches(d1, d2, ratio))
        if ratio > 1 and min(n1, n2) > 30:
            assert (got["rank"] > 1).any()


@pytest.mark.parametrize("case", ["random", "clusters", "big_norms", "odd_dim",
                                  "near_ties", "tiny", "mixed_magnitude", "dim65"])
def test_mfma_prefilter_equals_exhaustive_and_oracle(oracle, case, tmp_path):
    """Key sets large enough (>= 256) for the default producer, the MFMA
    prefilter with exact re-ranking (match_mfma.hip): the match lists must be
    the oracle's, entry by entry - and a child process with
    SARA_HIP_MATCH=exhaustive must return the same bytes.  Adversarial inputs:
    `clusters` puts hundreds of keys inside the prefilter's error guard of each
    other (slot overflow -> exhaustive fallback per query), `big_norms` has
    components up to 1e4 (large guard), `odd_dim` is not a multiple of four.
    Round 6 (the bf16 hi / lo bound assumes round-to-nearest f32
    accumulation in the matrix cores and no flushing of tiny lo halves):
    `near_ties` - every query has its 2nd..5th neighbours within a few ulp of
    each other (the order of near-tied 3rd / 4th neighbours decides the list);
    `tiny` - components around 1e-19, where x - hi underflows towards the
    subnormals; `mixed_magnitude` - rows of norm 1e4 next to rows of norm 1e-2
    in one tile (the bound scales with max |b|^2); `dim65` - one chunk of 64
    plus a single column."""
    import os
    import subprocess
    import sys
    rng = np.random.default_rng(len(case))
    dim = 128
    if case == "random":
        d1 = (rng.random((1500, dim), dtype=np.float32) * 255).astype(np.float32)
        d2 = (rng.random((1300, dim), dtype=np.float32) * 255).astype(np.float32)
        d2[:600] = d1[:600] + rng.normal(0, 2.0, (600, dim)).astype(np.float32)
        d2[600:620] = d1[700:720]                       # exact duplicates
    elif case == "clusters":
        centre = (rng.random((4, dim), dtype=np.float32) * 255).astype(np.float32)
        d1 = np.repeat(centre, 100, axis=0) + \
            rng.normal(0, 1e-3, (400, dim)).astype(np.float32)
        d2 = np.repeat(centre[::-1], 90, axis=0) + \
            rng.normal(0, 1e-3, (360, dim)).astype(np.float32)
    elif case == "big_norms":
        d1 = (rng.random((700, dim), dtype=np.float32) * 1e4).astype(np.float32)
        d2 = d1[::-1][:650] + rng.normal(0, 5.0, (650, dim)).astype(np.float32)
        d2 = d2.astype(np.float32)
    elif case == "near_ties":
        base = (rng.random((300, dim), dtype=np.float32) * 255).astype(np.float32)
        d1 = base.copy()
        # five copies of every key, each moved by about one ulp of its largest
        # components in a few places: distances to them differ in the last bits
        reps = []
        for k in range(5):
            c = base.copy()
            cols = rng.integers(0, dim, (300, 3))
            for j in range(3):
                c[np.arange(300), cols[:, j]] = np.nextafter(
                    c[np.arange(300), cols[:, j]], np.float32(1e9))
            reps.append(c)
        d2 = np.concatenate(reps).astype(np.float32)
        d2 = d2[rng.permutation(len(d2))]
    elif case == "tiny":
        d1 = (rng.random((500, dim), dtype=np.float32) * 1e-19).astype(np.float32)
        d2 = (d1[::-1][:450] * np.float32(1.001)).astype(np.float32)
        d2[:50] = (rng.random((50, dim), dtype=np.float32) * 1e-19).astype(np.float32)
    elif case == "mixed_magnitude":
        big = (rng.random((300, dim), dtype=np.float32) * 1e3).astype(np.float32)
        small = (rng.random((300, dim), dtype=np.float32) * 1e-3).astype(np.float32)
        d1 = np.concatenate([big, small])[rng.permutation(600)]
        d2 = np.concatenate([big[:250] + rng.normal(0, 0.5, (250, dim)).astype(np.float32),
                             small[:250] * np.float32(1.01),
                             small[250:] + np.float32(1e-6)]).astype(np.float32)
        d2 = d2[rng.permutation(len(d2))]
    elif case == "dim65":
        dim = 65
        d1 = (rng.random((450, dim), dtype=np.float32) * 255).astype(np.float32)
        d2 = (rng.random((380, dim), dtype=np.float32) * 255).astype(np.float32)
        d2[:200] = d1[100:300] + rng.normal(0, 1.0, (200, dim)).astype(np.float32)
    else:
        dim = 57
        d1 = rng.random((400, dim), dtype=np.float32)
        d2 = rng.random((300, dim), dtype=np.float32)
        d2[:100] = d1[50:150] + rng.normal(0, 1e-2, (100, dim)).astype(np.float32)
    np.save(tmp_path / "d1.npy", d1)
    np.save(tmp_path / "d2.npy", d2)
    ratios = (0.8, 1.0, 1.2)
    got = {r: sara_amd.match(d1, d2, r) for r in ratios}
    for r in ratios:
        assert_same(got[r], oracle.compute_matches(d1, d2, r))
    assert len(got[0.8]) > 0 or case in ("clusters", "near_ties")
    script = (
        "import sys, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "import sara_amd\n"
        "d1 = np.load(%r); d2 = np.load(%r)\n"
        "for r in %r:\n"
        "    np.save(%r %% r, sara_amd.match(d1, d2, r))\n"
        % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
           str(tmp_path / "d1.npy"), str(tmp_path / "d2.npy"), ratios,
           str(tmp_path / "ex_%s.npy")))
    env = dict(os.environ, SARA_HIP_MATCH="exhaustive")
    subprocess.run([sys.executable, "-c", script], check=True, env=env)
    for r in ratios:
        other = np.load(str(tmp_path / "ex_%s.npy") % r)
        assert other.tobytes() == got[r].tobytes(), (case, r)


def test_mfma_prefilter_self_matching(oracle):
    """The self-matching constructor on a set large enough for the prefilter
    (rank 0 = the key itself, top1 = 1)."""
    rng = np.random.default_rng(9)
    n = 900
    d = (rng.random((n, 128), dtype=np.float32) * 200).astype(np.float32)
    d[300:600] = d[:300] + rng.normal(0, 3.0, (300, 128)).astype(np.float32)
    reg = np.zeros(n, sara_amd.OEREGION_DTYPE)
    reg["coords"] = rng.random((n, 2), dtype=np.float32) * 2000
    reg["shape_matrix"] = np.array([0.04, 0, 0, 0.04], np.float32)
    reg["type"] = 5
    keys = sara_amd.KeypointList(reg, d, np.zeros((n, 2), np.int32))
    for ratio in (1.2, 1.6):
        got = sara_amd.AnnMatcher(keys, ratio).compute_matches()
        assert_same(got, oracle.compute_self_matches(d, reg, ratio))
        assert len(got) > 100


def test_sift_keypoints_of_shifted_frames(oracle):
    """The consumer's use: keypoints of a frame against those of the same scene
    shifted by a few pixels (match(), KeypointMatching.cpp:19-25)."""
    img = synth(360, 300, 21)
    a = np.ascontiguousarray(img[:280, :320])
    b = np.ascontiguousarray(img[8:288, 24:344])
    p = sara_amd.ImagePyramidParams(0, 6, num_octaves_max=3)
    ka = sara_amd.compute_sift_keypoints(a, p)
    kb = sara_amd.compute_sift_keypoints(b, p)
    got = sara_amd.match(ka, kb, 0.6)
    want = oracle.compute_matches(ka.descriptor_matrix, kb.descriptor_matrix, 0.6)
    assert_same(got, want)
    assert len(got) > 0.2 * min(len(ka), len(kb))
    # matched keypoints differ by the shift (24, 8)
    xa = ka.regions["coords"][got["x_index"]]
    xb = kb.regions["coords"][got["y_index"]]
    d = xa - xb
    good = (np.abs(d[:, 0] - 24) < 1.5) & (np.abs(d[:, 1] - 8) < 1.5)
    assert good.mean() > 0.9


def test_duplicates_and_ties(oracle):
    rng = np.random.default_rng(5)
    d1 = rng.random((20, 128), dtype=np.float32)
    d2 = np.concatenate([d1[:10], d1[:10], rng.random((15, 128), dtype=np.float32)])
    # exact duplicates: best == second best == 0 -> score 0 (d1 > 0 test fails)
    assert_same(sara_amd.match(d1, d2, 0.8), oracle.compute_matches(d1, d2, 0.8))


def test_error_behaviour():
    d = np.zeros((4, 128), np.float32)
    with pytest.raises(sara_amd.SaraHipError):
        sara_amd.match(np.zeros((0, 128), np.float32), d, 0.6)   # empty key set
    assert len(sara_amd.match(d, d, 1.2)) == 0    # best distance 0 -> radius 0
    with pytest.raises(sara_amd.SaraHipError):
        sara_amd.match(np.zeros((4, 200), np.float32),
                       np.zeros((4, 200), np.float32), 0.6)       # dim > 128


def test_default_ratio_on_sift_keypoints(oracle):
    """AnnMatcher{keys1, keys2} with the reference's default arguments
    (sift_ratio_thres = 1.2f): radius search, ranks 1..K, scores d_rank / d_best."""
    img = synth(360, 300, 22)
    a = np.ascontiguousarray(img[:280, :320])
    b = np.ascontiguousarray(img[8:288, 24:344])
    p = sara_amd.ImagePyramidParams(0, 6, num_octaves_max=3)
    ka = sara_amd.compute_sift_keypoints(a, p)
    kb = sara_amd.compute_sift_keypoints(b, p)
    got = sara_amd.AnnMatcher(ka, kb).compute_matches()
    want = oracle.compute_matches(ka.descriptor_matrix, kb.descriptor_matrix, 1.2)
    assert_same(got, want)
    assert got["rank"].max() > 1 and got["rank"].min() == 1
    assert np.all(got["score"] <= np.float32(1.2) * np.float32(1.2))
    # the ratio-0.6 matches are among the rank-1 matches - except those at
    # distance exactly 0 (the shift is a whole number of pixels, so interior
    # keypoints have bit-identical descriptors): radius = 0 * 1.44 = 0 and the
    # reference's radius search then returns nothing for that key (kept)
    strict = sara_amd.match(ka, kb, 0.6)
    r1 = {(int(m["x_index"]), int(m["y_index"])) for m in got if m["rank"] == 1}
    nz = {(int(m["x_index"]), int(m["y_index"])) for m in strict if m["score"] > 0}
    assert nz and nz <= r1


@pytest.mark.parametrize("ratio,metric,pixel", [(1.2, 0.5, 10.0), (1.5, 0.1, 2.0),
                                                (2.5, 0.5, 40.0), (1.0, 0.5, 10.0),
                                                (0.8, 0.5, 10.0)])
def test_self_matching_constructor(oracle, ratio, metric, pixel):
    """AnnMatcher{keys, ratio, min_max_metric_dist_thres, pixel_dist_thres}
    (AnnMatcher.cpp:199-215): rank 0 of each search is the key itself,
    KeyProximity (KeyProximity.cpp:17-30) drops neighbours that are too close,
    duplicates removed by value.  With ratio <= 1 the reference emits nothing."""
    img = synth(320, 280, 23)
    p = sara_amd.ImagePyramidParams(0, 6, num_octaves_max=3)
    keys = sara_amd.compute_sift_keypoints(img, p)
    assert len(keys) > 150
    got = sara_amd.AnnMatcher(keys, ratio, metric, pixel).compute_matches()
    want = oracle.compute_self_matches(keys.descriptor_matrix, keys.regions, ratio,
                                       metric, pixel)
    assert_same(got, want)
    if ratio <= 1:
        assert len(got) == 0
        return
    assert len(got) > 0
    assert np.all(got["x_index"] != got["y_index"])
    # nothing that KeyProximity calls redundant survives
    f = oracle.match_features(keys.regions)
    for m in got[:200]:
        assert not oracle.key_proximity(f[m["x_index"]], f[m["y_index"]], metric, pixel)


def test_self_matching_small_sets_and_duplicates(oracle):
    """Boundary cases of append_nearest_neighbors in self-matching mode
    (AnnMatcher.cpp:80-120): one key, two keys (score 1, no proximity test),
    three keys; and exact duplicates (rank 0 is then a twin of lower index)."""
    rng = np.random.default_rng(3)
    for n in (1, 2, 3, 5, 40):
        d = (rng.random((n, 128), dtype=np.float32) * 100).astype(np.float32)
        reg = np.zeros(n, sara_amd.OEREGION_DTYPE)
        reg["coords"] = rng.random((n, 2), dtype=np.float32) * 300
        reg["shape_matrix"] = np.array([0.25, 0, 0, 0.25], np.float32)
        reg["type"] = 5
        if n >= 5:
            d[3] = d[1]                      # exact duplicate descriptors
            reg["coords"][4] = reg["coords"][2] + 1   # too close in the image
        keys = sara_amd.KeypointList(reg, d, np.zeros((n, 2), np.int32))
        for ratio in (1.2, 3.0):
            got = sara_amd.AnnMatcher(keys, ratio).compute_matches()
            want = oracle.compute_self_matches(d, reg, ratio)
            assert_same(got, want)


def test_key_proximity_reference_case(oracle):
    """test_featurematching_key_proximity.cpp:26-37: f1 = (0, 0), f2 = (0, 0.1)
    with identity shape matrices are too close."""
    f1 = np.array([0, 0, 1, 0, 0, 1, 0, 11], np.float32)
    f2 = np.array([0, 0.1, 1 / 1.1 ** 2, 0, 0, 1 / 1.1 ** 2, 0, 11], np.float32)
    assert oracle.key_proximity(f1, f2)


def test_match_frames_on_device(oracle):
    """Frames of one batch matched where their descriptors are in HBM."""
    img = synth(360, 300, 21)
    frames = np.stack([img[:280, :320], img[8:288, 24:344], img[4:284, 10:330]])
    p = sara_amd.ImagePyramidParams(0, 6, num_octaves_max=3)
    with sara_amd.SiftContext(320, 280, 3, p) as ctx:
        ctx.detect(np.ascontiguousarray(frames))
        kl = ctx.keypoint_lists()
        for (i, j, ratio) in ((0, 1, 0.6), (2, 0, 0.6), (1, 1, 0.6), (0, 2, 1.2)):
            got = ctx.match_frames(i, j, ratio)
            want = oracle.compute_matches(kl[i].descriptor_matrix,
                                          kl[j].descriptor_matrix, ratio)
            assert_same(got, want)
            assert len(got) > 0


# ---- RootSIFT ---------------------------------
@pytest.mark.parametrize("n,dim", [(1, 128), (257, 128), (50, 6), (13, 200)])
def test_root_sift_operator_matches_oracle(oracle, n, dim):
    """RootSIFT.hpp:45-53: row /= L1 norm, then sqrt.  The device sums |h| in a
    different order than the oracle's left-to-right loop over 128 floats: rtol 4e-6."""
    rng = np.random.default_rng(n + dim)
    d = (rng.random((n, dim), dtype=np.float32) * 255).astype(np.float32)
    d[rng.random((n, dim)) < 0.3] = 0
    d[rng.random((n, dim)) < 0.1] *= -1   # Sara's descriptor has negative bins
    if n > 4:
        d[3] = 0                      # all-zero descriptor: left untouched
    got = sara_amd.root_sift(d)
    want = oracle.root_sift(d)
    assert np.allclose(got, want, rtol=4e-6, atol=0)
    nz = np.abs(d).sum(1) > 0
    assert np.allclose((got[nz].astype(np.float64) ** 2).sum(1), 1, atol=1e-5)
    assert np.array_equal(got[~nz], d[~nz])


def test_root_sift_option_of_the_pipeline(oracle):
    """SARA_HIP_OPT_ROOT_SIFT: the descriptor kernel emits RootSIFT rows; equal
    to post-processing the plain descriptors, everything else unchanged."""
    img = synth(320, 240, 5)
    p = sara_amd.ImagePyramidParams(0, 6, num_octaves_max=3)
    with sara_amd.SiftContext(320, 240, 1, p) as ctx:
        plain = ctx.detect(img).keypoint_lists()[0]
        ctx.set_option(sara_amd.capi.OPT_ROOT_SIFT, 1)
        root = ctx.detect(img).keypoint_lists()[0]
        ctx.set_option(sara_amd.capi.OPT_ROOT_SIFT, 0)
        again = ctx.detect(img).keypoint_lists()[0]
    assert len(plain) > 100
    assert plain.regions.tobytes() == root.regions.tobytes()
    assert np.array_equal(plain.descriptor_matrix, again.descriptor_matrix)
    want = oracle.root_sift(plain.descriptor_matrix)
    assert np.allclose(root.descriptor_matrix, want, rtol=4e-6, atol=0)
    assert np.allclose((root.descriptor_matrix.astype(np.float64) ** 2).sum(1), 1,
                       atol=1e-5)


# --------------------------------------------------------------------------- #
# Round 4: the GPU matcher against the reference's OWN FLANN.  The fixture
# tests/golden/flann_pins.npz holds what flann::Index<L2<float>> with
# LinearIndexParams (FLANN's exact index, the vendored header-only library under
# /root/reference/cpp/third-party/flann) returns when it is called as
# FeatureMatching/AnnMatcher.cpp:59-268 calls it; tests/test_oracle_flann_pins.py
# proves the oracle equal to it on the CPU.  Here the C-ABI entry points must
# return the same bytes: indices, ranks, directions, float scores.
# --------------------------------------------------------------------------- #
import hashlib
import os

from common import GOLDEN


def _sha(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


@pytest.fixture(scope="module")
def flann_pins():
    return np.load(os.path.join(GOLDEN, "flann_pins.npz"))


@pytest.fixture(scope="module")
def flann_pair(oracle, flann_pins):
    import sys
    sys.path.insert(0, GOLDEN)
    from make_flann_pins import pair_descriptors
    d1, d2 = pair_descriptors()
    assert _sha(d1, d2) == str(flann_pins["pair_sha256"])
    return d1, d2


@pytest.mark.parametrize("ratio", [0.6, 0.8, 1.0, 1.2])
def test_gpu_matcher_equals_annmatcher_on_flann_linear(flann_pair, flann_pins, ratio):
    d1, d2 = flann_pair
    got = sara_amd.AnnMatcher(d1, d2, ratio).compute_matches()
    want = flann_pins["linear_matches_%.1f" % ratio]
    assert len(got) == len(want) > 4000
    assert got.tobytes() == want.tobytes()


def test_gpu_self_matcher_equals_annmatcher_on_flann_linear(flann_pins):
    z = np.load(os.path.join(GOLDEN, "sunflower_crop.npz"))
    assert _sha(z["descriptors"], z["regions"]) == str(flann_pins["crop_sha256"])
    reg = np.ascontiguousarray(z["regions"]).view(sara_amd.OEREGION_DTYPE).reshape(-1)
    keys = sara_amd.KeypointList(reg, np.ascontiguousarray(z["descriptors"]),
                                 np.ascontiguousarray(z["scale_octave"]))
    got = sara_amd.AnnMatcher(keys, 1.2, 0.5, 10.0).compute_matches()
    want = flann_pins["linear_self_matches_crop"]
    assert len(got) == len(want) > 10000
    assert got.tobytes() == want.tobytes()


# ---- batched entry point (round 6) -------------------------------------------
@pytest.mark.parametrize("ratio", [0.6, 0.8, 1.0, 1.2])
def test_batched_matcher_equals_single_calls_on_the_flann_pair(flann_pair, flann_pins,
                                                               ratio):
    """sara_hip_match_descriptors_batch: a stream of pairs in one call, four
    searches in flight (ratios <= 1).  Every list is byte-identical to the
    single call's - and so to the reference's FLANN (tests/golden/flann_pins.npz).
    Ten pairs over four lanes: the same pair several times, the swapped pair,
    sub-sets of different sizes (the lanes' workspaces are re-used with growing
    and shrinking scratch)."""
    d1, d2 = flann_pair
    pairs = [(d1, d2), (d2, d1), (d1[:1000], d2[:3000]), (d1, d2), (d1[:70], d2[:50]),
             (d1[2000:], d2), (d1, d2), (d2[:129], d2[:129]), (d1, d2[::2]), (d1, d2)]
    got = sara_amd.match_pairs(pairs, ratio)
    assert len(got) == len(pairs)
    want = flann_pins["linear_matches_%.1f" % ratio]
    for k in (0, 3, 6, 9):
        assert got[k].tobytes() == want.tobytes(), k
    for k, (a, b) in enumerate(pairs):
        single = sara_amd.AnnMatcher(a, b, ratio).compute_matches()
        assert got[k].tobytes() == single.tobytes(), k
    assert sum(len(g) for g in got) > 20000


def test_batched_matcher_capacity_and_errors(flann_pair):
    from sara_amd import capi
    import ctypes as C
    d1, d2 = flann_pair
    lib = capi.load()
    arr = (capi.MatchPairStruct * 3)()
    for k in range(3):
        arr[k] = capi.MatchPairStruct(d1.ctypes.data, d2.ctypes.data, len(d1), len(d2))
    offsets = (C.c_int * 4)()
    out = np.zeros(100, sara_amd.MATCH_DTYPE)
    st = lib.sara_hip_match_descriptors_batch(arr, 3, 128, 0.6, 0, out.ctypes.data,
                                              100, offsets, 0)
    assert st == capi.CAPACITY_EXCEEDED
    single = sara_amd.match(d1, d2, 0.6)
    assert offsets[3] == 3 * len(single)          # the total needed
    # an empty key set fails the whole call before anything runs
    empty = np.zeros((0, 128), np.float32)
    with pytest.raises(sara_amd.SaraHipError):
        sara_amd.match_pairs([(d1, d2), (empty, d2)], 0.6)
    assert sara_amd.match_pairs([], 0.6) == []
    # and the next call is unaffected
    again = sara_amd.match_pairs([(d1, d2)], 0.6)
    assert again[0].tobytes() == single.tobytes()


def test_batched_match_of_consecutive_frames_on_device(oracle):
    """The consumer's loop (SfM: frame i against frame i + 1) on descriptors
    that never leave HBM."""
    img = synth(420, 300, 23)
    frames = np.stack([np.ascontiguousarray(img[:280, 6 * k:6 * k + 320])
                       for k in range(6)])
    p = sara_amd.ImagePyramidParams(0, 6, num_octaves_max=3)
    with sara_amd.SiftContext(320, 280, 6, p) as ctx:
        ctx.detect(frames)
        kl = ctx.keypoint_lists()
        pairs = [(k, k + 1) for k in range(5)] + [(5, 0)]
        got = ctx.match_frame_pairs(pairs, 0.6)
        for (i, j), g in zip(pairs, got):
            want = oracle.compute_matches(kl[i].descriptor_matrix,
                                          kl[j].descriptor_matrix, 0.6)
            assert_same(g, want)
            assert g.tobytes() == ctx.match_frames(i, j, 0.6).tobytes()
        assert min(len(g) for g in got[:5]) > 50


@pytest.mark.parametrize("ratio", [0.6, 0.8, 1.0])
def test_one_set_of_launches_for_a_batch_of_large_pairs(flann_pair, flann_pins, ratio):
    """Pairs whose sets all have >= 256 keys take the batched kernels: the pair
    is a grid dimension of every launch (one tile grid over all pairs, one
    read-back).  Different sizes in one batch (the grid covers the largest
    pair, the others' surplus workgroups leave), 40 pairs (two sets of
    launches), host and device descriptors."""
    d1, d2 = flann_pair
    pairs = [(d1, d2), (d2, d1), (d1[:1000], d2[:3000]), (d1[2000:], d2),
             (d2[:300], d1[:257]), (d1, d2[::2]), (d1[:256], d2[:256])]
    pairs = pairs + pairs[::-1] + pairs * 3 + pairs[:5]
    assert len(pairs) == 40
    got = sara_amd.match_pairs(pairs, ratio)
    want = flann_pins["linear_matches_%.1f" % ratio]
    assert got[0].tobytes() == want.tobytes()
    singles = {}
    for k, (a, b) in enumerate(pairs):
        key = (a.ctypes.data, b.ctypes.data, len(a), len(b))
        if key not in singles:
            singles[key] = sara_amd.AnnMatcher(a, b, ratio).compute_matches()
        assert got[k].tobytes() == singles[key].tobytes(), k
    # the same from descriptors that are already in HBM
    with sara_amd.DeviceArray(d1) as t1, sara_amd.DeviceArray(d2) as t2:
        raw = [(t1.ptr, len(d1), t2.ptr, len(d2)), (t2.ptr, len(d2), t1.ptr, len(d1)),
               (t1.ptr, 1000, t2.ptr, 3000)]
        dev = sara_amd._match_batch(raw, 128, ratio, 1, 0, 3 * (len(d1) + len(d2)))
    for k in range(3):
        assert dev[k].tobytes() == got[k].tobytes(), k
