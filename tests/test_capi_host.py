"""CPU-side checks of the C-ABI library: it loads, exports every symbol that
include/sara_hip_sift.h declares, its host-only arithmetic (parameter
schedule, Gaussian taps) agrees with the oracle, and without a GPU every
compute entry point fails loudly instead of falling back.  No kernels run.
"""
import ctypes as C
import os
import re

import numpy as np
import pytest

import sara_amd
from sara_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__
    if not os.path.exists(capi.LIB_PATH):
        __graft_entry__.build()
    return capi.load()


def test_header_and_binding_agree(lib):
    hdr = open(os.path.join(ROOT, "include", "sara_hip_sift.h")).read()
    declared = set(re.findall(r"\b(sara_hip_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"sara_hip_status", "sara_hip_stage", "sara_hip_sift"}
    assert declared == set(capi.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name


def test_pod_layouts():
    assert sara_amd.OEREGION_DTYPE.itemsize == 48
    assert C.sizeof(capi.PyramidParamsStruct) == 28
    assert C.sizeof(capi.SiftParamsStruct) == 44
    f = sara_amd.OEREGION_DTYPE.fields
    assert [f[n][1] for n in ("coords", "shape_matrix", "orientation",
                              "extremum_value", "type", "extremum_type")] == \
        [0, 16, 32, 36, 40, 41]


def test_defaults(lib):
    p = capi.SiftParamsStruct()
    lib.sara_hip_default_sift_params(C.byref(p))
    assert p.pyramid.first_octave_index == -1
    assert p.pyramid.scale_count_per_octave == 6
    assert p.pyramid.scale_geometric_factor == np.float32(2.0) ** np.float32(1 / 3)
    assert p.pyramid.image_padding_size == 1
    assert p.pyramid.scale_camera == 0.5
    assert p.pyramid.scale_initial == np.float32(1.6)
    assert p.pyramid.num_octaves_max == 2 ** 31 - 1
    assert (p.gauss_truncate, p.extremum_thres, p.edge_ratio_thres,
            p.extremum_refinement_iter) == (4.0, np.float32(0.01), 10.0, 5)
    # the Python-facing default first octave is +1 (pybind11 binding).
    assert sara_amd.ImagePyramidParams().first_octave_index == 1


@pytest.mark.parametrize("first,w,h,cap", [
    (0, 1920, 1080, 4), (0, 1920, 1080, 2 ** 31 - 1), (0, 3840, 2160, 5),
    (0, 1600, 1200, 4), (-1, 16, 16, 2 ** 31 - 1), (-1, 16, 16, 2),
    (1, 640, 480, 2 ** 31 - 1), (0, 11, 11, 2 ** 31 - 1), (0, 135, 67, 3),
])
def test_octave_schedule_matches_oracle(lib, oracle, first, w, h, cap):
    hp = sara_amd.ImagePyramidParams(first, 6, num_octaves_max=cap)
    rp = oracle.PyramidParams(first, 6, None, 1, 0.5, 1.6, cap)
    img = np.zeros((h, w), np.float32)
    r = oracle.RefSift(img, rp, pyramid_only=True)
    assert hp.octave_count(w, h) == r.octave_count
    for o in range(r.octave_count):
        assert hp.octave_info(w, h, o) == r.octave_info(o)
    with pytest.raises(sara_amd.SaraHipError) as e:
        hp.octave_info(w, h, r.octave_count)
    assert e.value.status == capi.OUT_OF_RANGE


def test_1080p_octave_sizes(lib):
    hp = sara_amd.ImagePyramidParams(0, num_octaves_max=4)
    assert [hp.octave_info(1920, 1080, o)[:2] for o in range(4)] == \
        [(1920, 1080), (960, 540), (480, 270), (240, 135)]
    assert sara_amd.ImagePyramidParams(0).octave_count(1920, 1080) == 9


@pytest.mark.parametrize("sigma,trunc", [
    (1.0, 1.0), (1.0, 4.0), (1.2262735, 4.0), (1.5450078, 4.0), (1.946588, 4.0),
    (2.452547, 4.0), (3.0900156, 4.0), (1.5198685, 4.0), (0.05, 4.0), (7.9, 4.0),
])
def test_gaussian_taps_match_oracle(lib, oracle, sigma, trunc):
    a = sara_amd.make_gaussian_kernel(sigma, trunc)
    b = oracle.make_gaussian_kernel(sigma, trunc)
    assert a.shape == b.shape
    assert np.array_equal(a, b)


@pytest.mark.parametrize("name,arith", [("eigen34_sse", capi.TAPS_EIGEN34_SSE2),
                                        ("eigen33_sse", capi.TAPS_EIGEN33_SSE2),
                                        ("expf_serial", capi.TAPS_LIBM_SERIAL)])
def test_gaussian_taps_match_oracle_under_each_arithmetic(lib, oracle, name, arith):
    """SARA_HIP_OPT_TAP_ARITHMETIC: the product's host code and the oracle's
    tap variants (oracle/sift_ref.hpp kTaps*) are the same floats."""
    sigmas = [1.51986849, 1.22627354, 1.54500782, 1.94658804, 2.45254707,
              3.09001565, 1.24899971, 0.3, 0.8, 5.3, 11.0]
    differs = asym = 0
    for sigma in sigmas:
        a = sara_amd.make_gaussian_kernel(sigma, 4.0, arith)
        with oracle.tap_variant(name):
            b = oracle.make_gaussian_kernel(sigma, 4.0)
        assert np.array_equal(a.view(np.int32), b.view(np.int32)), sigma
        base = sara_amd.make_gaussian_kernel(sigma, 4.0)
        differs += not np.array_equal(a, base)
        asym += not np.array_equal(a, a[::-1])
        # the reference's own acceptance (test_imageprocessing_linear_filtering
        # .cpp:136-187): 1e-5 in L2 - every arithmetic is far inside it
        assert np.linalg.norm(a.astype(np.float64) - base) < 1e-6
    if arith != capi.TAPS_LIBM_SERIAL:
        assert differs >= 4      # the choice is visible in the last ulp ...
    if arith == capi.TAPS_EIGEN34_SSE2:
        assert asym >= 2         # ... and the packet / scalar split breaks symmetry


def test_no_gpu_fails_loudly(lib):
    if lib.sara_hip_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(sara_amd.SaraHipError) as e:
        sara_amd.SiftContext(64, 64)
    assert e.value.status == capi.NO_DEVICE
    with pytest.raises(sara_amd.SaraHipError) as e:
        sara_amd.compute_sift_keypoints(np.zeros((32, 32), np.float32))
    assert e.value.status == capi.NO_DEVICE
    with pytest.raises(sara_amd.SaraHipError) as e:
        sara_amd.apply_gaussian_filter(np.zeros((8, 8), np.float32), 1.0)
    assert e.value.status == capi.NO_DEVICE
    # the raw C entry point reports it too
    h = C.c_void_p()
    p = capi.SiftParamsStruct()
    lib.sara_hip_default_sift_params(C.byref(p))
    st = lib.sara_hip_sift_create(C.byref(p), 64, 64, 1, 0, 0, C.byref(h))
    assert st == capi.NO_DEVICE and not h.value
    assert b"no CPU fallback" in lib.sara_hip_last_error()


def test_invalid_params_rejected_before_device(lib):
    p = capi.SiftParamsStruct()
    lib.sara_hip_default_sift_params(C.byref(p))
    p.pyramid.scale_count_per_octave = 3
    h = C.c_void_p()
    st = lib.sara_hip_sift_create(C.byref(p), 64, 64, 1, 0, 0, C.byref(h))
    assert st == capi.INVALID_PARAMS
    assert b"4 scales per octave" in lib.sara_hip_last_error()
    with pytest.raises(RuntimeError):
        sara_amd.ComputeDoGExtrema(sara_amd.ImagePyramidParams(0, 3))
    # argument checks of the stand-alone operators come before any device call
    assert lib.sara_hip_root_sift(None, 4, 128, 0, 0) == capi.INVALID_PARAMS
    buf = (C.c_float * 4)()
    assert lib.sara_hip_root_sift(buf, 1, 0, 0, 0) == capi.INVALID_PARAMS
    assert lib.sara_hip_root_sift(buf, 0, 4, 0, 0) == capi.OK      # nothing to do
    assert lib.sara_hip_selfcheck_device_math(None, 0) == capi.INVALID_PARAMS
    assert lib.sara_hip_sift_set_option(None, capi.OPT_ROOT_SIFT, 1) == capi.INVALID_PARAMS


def test_product_does_not_reference_oracle():
    """The product tree must never import, link or call the oracle."""
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "sara_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h", "Makefile")):
                txt = open(os.path.join(base, f), errors="ignore").read()
                if re.search(r"refbind|sift_ref|libsift_ref|oracle/", txt):
                    bad.append(f)
    assert not bad, bad


def test_shard_ranges_cover_the_frames_in_order():
    """sara_hip_shard_range: contiguous blocks, frame f on rank floor(f * W / N),
    empty shards when there are fewer frames than ranks - the same split as
    sara_amd.distributed.shard_range (the torch.distributed variant)."""
    from sara_amd import distributed as sd
    for n, w in ((512, 8), (64, 1), (10, 4), (3, 8), (0, 4), (7, 7), (1000, 6)):
        at = 0
        for r in range(w):
            lo, hi = sd.shard_range_native(n, w, r)
            assert (lo, hi) == sd.shard_range(n, w, r)
            assert lo == at and hi >= lo
            for f in range(lo, hi):
                assert (f * w) // n == r
            at = hi
        assert at == n


def test_multi_gpu_entry_points_fail_loudly_without_a_gpu():
    from sara_amd import capi
    import ctypes as C
    lib = capi.load()
    if lib.sara_hip_device_count() > 0:
        return
    sp = capi.SiftParamsStruct()
    lib.sara_hip_default_sift_params(C.byref(sp))
    g = C.c_void_p()
    st = lib.sara_hip_sift_group_create(C.byref(sp), 64, 64, 1, 0, 1, None, C.byref(g))
    assert st in (capi.NO_DEVICE, capi.RCCL_ERROR) and not g.value
