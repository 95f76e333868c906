"""-m gpu: the drop-in has no keypoint limit, like the reference.

Sara keeps its extrema in std::vectors (`reserve(10000)` per scale, then
`push_back`: FeatureDetectors/RefineExtremum.cpp:496-514), so
compute_sift_keypoints() (FeatureDetectors/SIFT.cpp:27-108) returns whatever
an image produces.  A context's lists in HBM have a capacity (default
w * h / 128 keypoints and four times as many classified sites per frame); the
free functions grow the lists and run the frame again when it overflows them
(sara_hip_sift_capacity / sara_hip_sift_reserve), and keep the grown context.

Images: a 10-pixel grid of 3 x 3 dots (a calibration target: 26 580 extrema /
58 478 keypoints at 1080p against a default capacity of 16 200), 4 x 4 blocks
of binary noise (extrema overflow, then keypoints) and noisy diagonal stripes
(more classified sites than 4 x a tiny capacity, no extremum).
Bars as in test_gpu_pipeline.py.
"""
import time

import numpy as np
import pytest

import common
import sara_amd
from sara_amd import capi
from test_gpu_pipeline import (DESC_ATOL, SHAPE_RTOL, THETA_ATOL, hip_params,
                               ref_params)

pytestmark = pytest.mark.gpu


def blocky_noise(w, h, seed=11, block=4):
    rng = np.random.default_rng(seed)
    b = (rng.random((h // block, w // block)) > 0.5).astype(np.float32)
    return np.kron(b, np.ones((block, block), np.float32))


def assert_keys_equal_oracle(keys, ref):
    rk, rso, rdesc = ref.keypoints()
    assert len(keys) == len(rk), (len(keys), len(rk))
    assert np.array_equal(keys.scale_octave, rso)
    common.assert_regions_equal(keys.regions, rk, rtol_shape=SHAPE_RTOL,
                                atol_theta=THETA_ATOL)
    assert np.max(np.abs(keys.descriptor_matrix - rdesc)) <= DESC_ATOL


@pytest.fixture()
def fresh_cache():
    sara_amd.clear_context_cache()
    yield
    sara_amd.clear_context_cache()


def cached_capacity():
    ctx = sara_amd._CONTEXTS.entries[0][1]
    return ctx.capacity()[0]


def test_free_function_on_a_dot_grid_1080p(oracle, fresh_cache):
    """The free function, no capacity argument, on a frame with 3.6 times the
    keypoints the default lists hold: equal to the oracle; the second call
    finds the grown context and costs what a call costs."""
    w, h = 1920, 1080
    img = common.dot_grid(w, h)
    ref = oracle.RefSift(img, ref_params(oracle, 0, 4))
    n_ref = len(ref.keypoints()[0])
    assert n_ref > w * h // 128, "the image must overflow the default lists"
    keys = sara_amd.compute_sift_keypoints(img, hip_params(0, 4))
    assert_keys_equal_oracle(keys, ref)
    cap = cached_capacity()
    assert cap >= n_ref
    # steady state: same context, no further growth, identical results
    t = []
    for _ in range(5):
        t0 = time.perf_counter()
        again = sara_amd.compute_sift_keypoints(img, hip_params(0, 4))
        t.append(time.perf_counter() - t0)
    assert cached_capacity() == cap
    assert again.regions.tobytes() == keys.regions.tobytes()
    assert again.descriptor_matrix.tobytes() == keys.descriptor_matrix.tobytes()
    print("dot grid 1080p: %d keypoints, capacity %d -> %d, steady call %.2f ms"
          % (n_ref, w * h // 128, cap, 1e3 * min(t)))
    # an ordinary frame through the grown context is still an ordinary frame
    from sara_amd.synth import synth
    img2 = synth(w, h, 1234)
    ref2 = oracle.RefSift(img2, ref_params(oracle, 0, 4))
    assert_keys_equal_oracle(sara_amd.compute_sift_keypoints(img2, hip_params(0, 4)),
                             ref2)
    assert cached_capacity() == cap


def test_free_function_on_blocky_noise(oracle, fresh_cache):
    """4 x 4 blocks of binary noise: 3 335 extrema / 4 505 keypoints at
    640 x 480 against a default capacity of 2 400 - the extremum list overflows
    first and starves the keypoint list, so the first `required` is too low
    for the keypoints and the loop takes what it needs."""
    w, h = 640, 480
    img = blocky_noise(w, h)
    ref = oracle.RefSift(img, ref_params(oracle, 0, 4))
    with sara_amd.SiftContext(w, h, 1, hip_params(0, 4)) as ctx:
        ctx.detect(img)
        with pytest.raises(capi.SaraHipError) as e:
            ctx.counts()
        assert e.value.status == capi.CAPACITY_EXCEEDED
        cap, need = ctx.capacity()
        assert cap == w * h // 128 and need > cap
    keys = sara_amd.compute_sift_keypoints(img, hip_params(0, 4))
    assert len(keys) > w * h // 128
    assert_keys_equal_oracle(keys, ref)


def test_site_list_overflow_alone_is_reported_and_grown(oracle):
    """The second limit: classified sites (4 x max_keypoints per frame).
    Diagonal stripes under noise give sites that the edge test rejects - more
    sites than 4 x a (deliberately tiny) capacity, no extremum at all: counts()
    reports it, `required` = sites / 4, and the grown context is exact."""
    w, h = 640, 480
    rng = np.random.default_rng(11)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    img = (0.5 + 0.5 * np.sin((xx + yy) * 0.5) +
           0.05 * rng.standard_normal((h, w))).astype(np.float32)
    ref = oracle.RefSift(img, ref_params(oracle, 0, 4))
    n_ex = len(ref.extrema()[0])
    with sara_amd.SiftContext(w, h, 1, hip_params(0, 4), max_keypoints=4) as ctx:
        ctx.detect(img)
        with pytest.raises(capi.SaraHipError) as e:
            ctx.counts()
        assert e.value.status == capi.CAPACITY_EXCEEDED
        assert "classified sites" in str(e.value)
        cap, need = ctx.capacity()
        assert cap == 4 and need > 4 and need > n_ex
        ctx.grow_for_last_batch()
        assert ctx.capacity()[0] == 2 * need
        ctx.detect(img)
        c, total = ctx.counts()
        rk = ref.keypoints()[0]
        assert total == len(rk)
        ec, ereg, _ = ctx.extrema()
        assert int(ec[0]) == n_ex


@pytest.mark.parametrize("kind", ["dots", "noise"])
def test_compute_dog_extrema_grows(oracle, kind):
    """ComputeDoGExtrema::operator() (FeatureDetectors/DoG.cpp:23-87)."""
    w, h = 800, 600
    img = common.dot_grid(w, h) if kind == "dots" else blocky_noise(w, h)
    ref = oracle.RefSift(img, ref_params(oracle, 0, 4))
    rreg, rxyso = ref.extrema()
    dog = sara_amd.ComputeDoGExtrema(hip_params(0, 4), img_padding_sz=5)
    regions, so = dog(img)
    assert len(regions) == len(rreg)
    common.assert_regions_equal(regions, rreg, rtol_shape=SHAPE_RTOL)
    assert np.array_equal(so, rxyso[:, 2:4])
    # the functor keeps its (grown) context for the next frame of that size
    regions2, _ = dog(img)
    assert regions2.tobytes() == regions.tobytes()
    assert np.array_equal(dog.gaussians(1, 0), ref.gaussian(1, 0))


def test_reserve_semantics(oracle):
    """C-ABI: reserve() never shrinks, refuses while a ticket is pending,
    forgets the last result, and a grown context equals one created large."""
    from sara_amd.synth import synth
    w, h = 640, 480
    img = synth(w, h, 5)
    with sara_amd.SiftContext(w, h, 2, hip_params(0, 4)) as ctx:
        cap0, _ = ctx.capacity()
        assert cap0 == w * h // 128
        ctx.reserve(100)
        assert ctx.capacity()[0] == cap0
        # both result slots exist before the growth
        t0 = ctx.submit(img)
        t1 = ctx.submit(img)
        with pytest.raises(capi.SaraHipError) as e:
            ctx.reserve(3 * cap0)
        assert e.value.status == capi.NOT_READY
        first = ctx.collect(t0, copy=True)  # byte-wise (padding included)
        ctx.collect(t1)
        assert len(first[1]) <= ctx.capacity()[1] <= cap0  # follows the batch
        ctx.reserve(3 * cap0)
        assert ctx.capacity()[0] == 3 * cap0
        with pytest.raises(capi.SaraHipError) as e:
            ctx.counts()
        assert e.value.status == capi.NOT_READY
        # graph replay (batch 1) and both slots after the growth
        for _ in range(3):
            t = ctx.submit(img)
            off, reg, desc, so = ctx.collect(t)
            assert reg.tobytes() == first[1].tobytes()
            assert desc.tobytes() == first[2].tobytes()
            assert so.tobytes() == first[3].tobytes()
        # both frames of a full batch
        ctx.detect(np.stack([img, img]))
        c, reg, desc, so = ctx.fetch()
        assert c[0] == c[1] == len(first[1])
        assert reg[:c[0]].tobytes() == first[1].tobytes()
        assert reg[c[0]:].tobytes() == first[1].tobytes()
    with sara_amd.SiftContext(w, h, 1, hip_params(0, 4), max_keypoints=3 * cap0) as big:
        big.detect(img)
        _, reg, desc, so = big.fetch()
        assert reg.tobytes() == first[1].tobytes()
        assert desc.tobytes() == first[2].tobytes()


def test_reserve_that_cannot_fit_keeps_the_context_usable():
    from sara_amd.synth import synth
    w, h = 320, 240
    img = synth(w, h, 5)
    with sara_amd.SiftContext(w, h, 1, hip_params(0, 3)) as ctx:
        ctx.detect(img)
        n0 = ctx.counts()[1]
        with pytest.raises(capi.SaraHipError) as e:
            ctx.reserve(2 ** 28)  # ~330 GB of lists: more than the HBM
        assert e.value.status == capi.CAPACITY_EXCEEDED
        ctx.detect(img)
        assert ctx.counts()[1] == n0
