"""ctypes binding of the C-ABI in include/sara_hip_sift.h.

Loads sara_amd/lib/libsara_hip_sift.so (built in-tree by
``__graft_entry__.build()`` / ``make -C sara_amd/csrc``).  There is no CPU
fallback: a missing library or a missing GPU raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SARA_HIP_SIFT_LIB: A/B a differently built library (tools/ab_build.sh).
LIB_PATH = os.environ.get("SARA_HIP_SIFT_LIB") or os.path.join(
    _HERE, "lib", "libsara_hip_sift.so")

# status codes (sara_hip_status)
OK, INVALID_PARAMS, SIZE_MISMATCH, OUT_OF_RANGE, CAPACITY_EXCEEDED, \
    RUNTIME_ERROR, NO_DEVICE, NOT_READY, RCCL_ERROR = range(9)

# stages (sara_hip_stage)
STAGE_PYRAMID, STAGE_EXTREMA, STAGE_GRADIENT, STAGE_ORIENTATION, \
    STAGE_DESCRIPTOR = 1, 2, 3, 4, 5

TIME_NAMES = ["upload", "pyramid", "extrema", "gradient", "orientation",
              "descriptor", "total"]

OPT_ALL_GRADIENT_SCALES = 1
OPT_STAGE_TIMERS = 2
OPT_ROOT_SIFT = 3
OPT_SIGNED_EXTREMUM_TYPE = 4
OPT_DOWNSCALE_AT_DOUBLE_SIGMA = 5
OPT_FMA_BLUR = 6
OPT_SINGLE_STREAM = 7
OPT_LAUNCH_TIMERS = 8
OPT_TAP_ARITHMETIC = 9
OPT_KERNEL_SELECTION = 10
OPT_TILE_GEOMETRY = 11
OPT_MARCH_WAVES = 12
OPT_GRAPH_REPLAY = 13
OPT_MARCH2_WAVES = 14
# values of OPT_KERNEL_SELECTION (sara_hip_sift.h SARA_HIP_SELECT_*)
SELECT_ENVIRONMENT = 0
SELECT_SHIPPED = 1
SELECT_FORCED_MARCH = 2
SELECT_TILED = 3
SELECT_TILED_BLUR = 4
# values of OPT_TAP_ARITHMETIC (sara_hip_sift.h SARA_HIP_TAPS_*)
TAPS_LIBM_SERIAL = 0
TAPS_EIGEN34_SSE2 = 1
TAPS_EIGEN33_SSE2 = 2

LAUNCH_TIME_DTYPE = np.dtype([("octave", "<i4"), ("scale", "<i4"), ("taps", "<i4"),
                              ("_pad", "<i4"), ("pixels", "<i8"), ("ms", "<f4"),
                              ("_pad2", "<f4")])

#: numpy view of sara_oeregion (48 bytes, Features/Feature.hpp:155-177).
class MatchPairStruct(C.Structure):
    """sara_match_pair: one pair of sara_hip_match_descriptors_batch."""
    _fields_ = [("desc1", C.c_void_p), ("desc2", C.c_void_p),
                ("n1", C.c_int32), ("n2", C.c_int32)]


MATCH_DTYPE = np.dtype([("x_index", "<i4"), ("y_index", "<i4"), ("score", "<f4"),
                        ("rank", "<i4"), ("direction", "<i4")])
OEREGION_DTYPE = np.dtype(
    {
        "names": ["coords", "shape_matrix", "orientation", "extremum_value",
                  "type", "extremum_type"],
        "formats": [("<f4", 2), ("<f4", 4), "<f4", "<f4", "u1", "i1"],
        "offsets": [0, 16, 32, 36, 40, 41],
        "itemsize": 48,
    }
)


class PyramidParamsStruct(C.Structure):
    _fields_ = [
        ("first_octave_index", C.c_int32),
        ("scale_count_per_octave", C.c_int32),
        ("scale_geometric_factor", C.c_float),
        ("image_padding_size", C.c_int32),
        ("scale_camera", C.c_float),
        ("scale_initial", C.c_float),
        ("num_octaves_max", C.c_int32),
    ]


class SiftParamsStruct(C.Structure):
    _fields_ = [
        ("pyramid", PyramidParamsStruct),
        ("gauss_truncate", C.c_float),
        ("extremum_thres", C.c_float),
        ("edge_ratio_thres", C.c_float),
        ("extremum_refinement_iter", C.c_int32),
    ]


#: every entry point include/sara_hip_sift.h declares.
EXPORTS = [
    "sara_hip_last_error", "sara_hip_version", "sara_hip_device_count",
    "sara_hip_default_pyramid_params", "sara_hip_default_sift_params",
    "sara_hip_pyramid_octave_count", "sara_hip_pyramid_octave_info",
    "sara_hip_make_gaussian_kernel", "sara_hip_make_gaussian_kernel_with",
    "sara_hip_sift_create",
    "sara_hip_sift_create_dog", "sara_hip_sift_destroy", "sara_hip_sift_detect",
    "sara_hip_sift_synchronize", "sara_hip_sift_counts", "sara_hip_sift_fetch",
    "sara_hip_sift_device_results", "sara_hip_sift_octave_count",
    "sara_hip_sift_octave_info", "sara_hip_sift_copy_gaussian",
    "sara_hip_sift_copy_dog", "sara_hip_sift_copy_gradient",
    "sara_hip_sift_extrema_counts", "sara_hip_sift_fetch_extrema",
    "sara_hip_sift_stage_times", "sara_hip_sift_set_option",
    "sara_hip_apply_gaussian_filter", "sara_hip_scale", "sara_hip_enlarge",
    "sara_hip_subtract", "sara_hip_gradient_polar_coordinates",
    "sara_hip_scale_space_dog_extremum_map", "sara_hip_selfcheck_atan2f",
    "sara_hip_sift_detect_u8", "sara_hip_from_rgb8_to_gray32f",
    "sara_hip_from_gray8_to_gray32f", "sara_hip_match_descriptors",
    "sara_hip_match_descriptors_batch",
    "sara_hip_sift_stage", "sara_hip_sift_detect_staged", "sara_hip_root_sift",
    "sara_hip_selfcheck_device_math", "sara_hip_selfcheck_sincos",
    "sara_hip_selfcheck_definiteness", "sara_hip_selfcheck_orientation_bins",
    "sara_hip_sift_submit", "sara_hip_sift_submit_staged", "sara_hip_sift_collect",
    "sara_hip_shard_range", "sara_hip_copy_to_host", "sara_hip_comm_unique_id", "sara_hip_comm_create",
    "sara_hip_comm_gather", "sara_hip_comm_destroy",
    "sara_hip_sift_group_create", "sara_hip_sift_group_size",
    "sara_hip_sift_group_context", "sara_hip_sift_group_detect",
    "sara_hip_sift_group_gather", "sara_hip_sift_group_destroy",
    "sara_hip_sift_ticket_counts", "sara_hip_sift_collect_into",
    "sara_hip_host_register", "sara_hip_host_unregister",
    "sara_hip_host_alloc", "sara_hip_host_free",
    "sara_hip_copy_to_device", "sara_hip_device_alloc", "sara_hip_device_free",
    "sara_hip_comm_transport", "sara_hip_sift_group_collect_host",
    "sara_hip_sift_group_transport", "sara_hip_self_match_descriptors",
    "sara_hip_match_release_workspace", "sara_hip_sift_pyramid_launches",
    "sara_hip_comm_size", "sara_hip_rccl_version",
    "sara_hip_sift_capacity", "sara_hip_sift_reserve",
]

_f32p = C.POINTER(C.c_float)
_i32p = C.POINTER(C.c_int32)
_vp = C.c_void_p


class SaraHipError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("sara_hip status %d: %s" % (status, message))
        self.status = status


def _declare(lib):
    lib.sara_hip_last_error.restype = C.c_char_p
    lib.sara_hip_version.restype = C.c_int
    lib.sara_hip_device_count.restype = C.c_int
    lib.sara_hip_default_pyramid_params.argtypes = [C.POINTER(PyramidParamsStruct)]
    lib.sara_hip_default_pyramid_params.restype = None
    lib.sara_hip_default_sift_params.argtypes = [C.POINTER(SiftParamsStruct)]
    lib.sara_hip_default_sift_params.restype = None
    lib.sara_hip_pyramid_octave_count.argtypes = [C.POINTER(PyramidParamsStruct),
                                                  C.c_int, C.c_int]
    lib.sara_hip_pyramid_octave_info.argtypes = [C.POINTER(PyramidParamsStruct),
                                                 C.c_int, C.c_int, C.c_int,
                                                 C.POINTER(C.c_int),
                                                 C.POINTER(C.c_int), _f32p]
    lib.sara_hip_make_gaussian_kernel.argtypes = [C.c_float, C.c_float, _f32p,
                                                  C.c_int]
    lib.sara_hip_make_gaussian_kernel_with.argtypes = [C.c_int, C.c_float,
                                                       C.c_float, _f32p, C.c_int]
    lib.sara_hip_sift_create.argtypes = [C.POINTER(SiftParamsStruct), C.c_int,
                                         C.c_int, C.c_int, C.c_int, C.c_int,
                                         C.POINTER(_vp)]
    lib.sara_hip_sift_create_dog.argtypes = [C.POINTER(PyramidParamsStruct),
                                             C.c_float, C.c_float, C.c_float,
                                             C.c_int, C.c_int, C.c_int, C.c_int,
                                             C.c_int, C.c_int, C.c_int,
                                             C.POINTER(_vp)]
    lib.sara_hip_sift_destroy.argtypes = [_vp]
    lib.sara_hip_sift_detect.argtypes = [_vp, _vp, C.c_size_t, C.c_int, C.c_int,
                                         C.c_int, C.c_int, C.c_int, _vp]
    lib.sara_hip_sift_synchronize.argtypes = [_vp]
    lib.sara_hip_sift_counts.argtypes = [_vp, _i32p, _i32p]
    lib.sara_hip_sift_capacity.argtypes = [_vp, C.POINTER(C.c_int),
                                           C.POINTER(C.c_int)]
    lib.sara_hip_sift_reserve.argtypes = [_vp, C.c_int]
    lib.sara_hip_sift_fetch.argtypes = [_vp, _vp, _vp, _vp, C.c_int]
    lib.sara_hip_sift_device_results.argtypes = [_vp, C.POINTER(_vp),
                                                 C.POINTER(_vp), C.POINTER(_vp),
                                                 C.POINTER(_vp)]
    lib.sara_hip_sift_octave_count.argtypes = [_vp]
    lib.sara_hip_sift_octave_info.argtypes = [_vp, C.c_int, C.POINTER(C.c_int),
                                              C.POINTER(C.c_int), _f32p]
    for name in ("sara_hip_sift_copy_gaussian", "sara_hip_sift_copy_dog",
                 "sara_hip_sift_copy_gradient"):
        getattr(lib, name).argtypes = [_vp, C.c_int, C.c_int, C.c_int, _f32p]
    lib.sara_hip_sift_extrema_counts.argtypes = [_vp, _i32p, _i32p]
    lib.sara_hip_sift_fetch_extrema.argtypes = [_vp, _vp, _i32p]
    lib.sara_hip_sift_stage_times.argtypes = [_vp, _f32p]
    lib.sara_hip_sift_set_option.argtypes = [_vp, C.c_int, C.c_int]
    lib.sara_hip_sift_pyramid_launches.argtypes = [_vp, _vp, C.c_int,
                                                   C.POINTER(C.c_int)]
    lib.sara_hip_apply_gaussian_filter.argtypes = [_f32p, _f32p, C.c_int,
                                                   C.c_int, C.c_float,
                                                   C.c_float, C.c_int]
    lib.sara_hip_scale.argtypes = [_f32p, C.c_int, C.c_int, _f32p, C.c_int,
                                   C.c_int, C.c_int]
    lib.sara_hip_enlarge.argtypes = [_f32p, C.c_int, C.c_int, _f32p, C.c_int,
                                     C.c_int, C.c_int]
    lib.sara_hip_subtract.argtypes = [_f32p, _f32p, _f32p, C.c_size_t, C.c_int]
    lib.sara_hip_gradient_polar_coordinates.argtypes = [_f32p, C.c_int, C.c_int,
                                                        _f32p, C.c_int]
    lib.sara_hip_scale_space_dog_extremum_map.argtypes = [
        _f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int,
        C.POINTER(C.c_int8), C.c_int]
    lib.sara_hip_sift_detect_u8.argtypes = [_vp, _vp, C.c_size_t, C.c_int,
                                            C.c_int, C.c_int, C.c_int, C.c_int,
                                            C.c_int, _vp]
    for name in ("sara_hip_from_rgb8_to_gray32f",
                 "sara_hip_from_gray8_to_gray32f"):
        getattr(lib, name).argtypes = [_vp, _f32p, C.c_int, C.c_int, C.c_int]
    lib.sara_hip_sift_stage.argtypes = [_vp, _vp, C.c_size_t, C.c_int, C.c_int,
                                        C.c_int, C.c_int]
    lib.sara_hip_sift_detect_staged.argtypes = [_vp, C.c_int, _vp]
    lib.sara_hip_match_descriptors.argtypes = [
        _vp, C.c_int, _vp, C.c_int, C.c_int, C.c_float, C.c_int, _vp, C.c_int,
        C.POINTER(C.c_int), C.c_int]
    lib.sara_hip_match_descriptors_batch.argtypes = [
        C.POINTER(MatchPairStruct), C.c_int, C.c_int, C.c_float, C.c_int, _vp,
        C.c_int, C.POINTER(C.c_int), C.c_int]
    lib.sara_hip_self_match_descriptors.argtypes = [
        _vp, _vp, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_int, _vp,
        C.c_int, C.POINTER(C.c_int), C.c_int]
    lib.sara_hip_match_release_workspace.argtypes = [C.c_int]
    lib.sara_hip_root_sift.argtypes = [_vp, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.sara_hip_selfcheck_atan2f.argtypes = [_f32p, _f32p, _f32p, C.c_size_t]
    lib.sara_hip_selfcheck_atan2f.restype = None
    lib.sara_hip_selfcheck_device_math.argtypes = [
        C.POINTER(C.c_ulonglong), C.c_int]
    lib.sara_hip_sift_submit.argtypes = [_vp, _vp, C.c_size_t, C.c_int, C.c_int,
                                         C.c_int, C.c_int, C.c_int, C.c_int,
                                         C.POINTER(C.c_int)]
    lib.sara_hip_sift_submit_staged.argtypes = [_vp, C.c_int, C.POINTER(C.c_int)]
    lib.sara_hip_sift_collect.argtypes = [_vp, C.c_int, C.POINTER(_vp),
                                          C.POINTER(_vp), C.POINTER(_vp),
                                          C.POINTER(_vp), C.POINTER(C.c_int)]
    lib.sara_hip_shard_range.argtypes = [C.c_int, C.c_int, C.c_int,
                                         C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.sara_hip_shard_range.restype = None
    lib.sara_hip_copy_to_host.argtypes = [_vp, _vp, C.c_size_t, C.c_int]
    lib.sara_hip_comm_unique_id.argtypes = [_vp]
    lib.sara_hip_comm_create.argtypes = [_vp, _vp, C.c_int, C.c_int, C.c_int,
                                         C.POINTER(_vp)]
    lib.sara_hip_comm_gather.argtypes = [_vp, C.c_int, C.c_int, C.c_int, _vp,
                                         C.POINTER(_vp), C.POINTER(_vp),
                                         C.POINTER(_vp), C.POINTER(C.c_int)]
    lib.sara_hip_comm_destroy.argtypes = [_vp]
    lib.sara_hip_sift_group_create.argtypes = [_vp, C.c_int, C.c_int, C.c_int,
                                               C.c_int, C.c_int, _vp,
                                               C.POINTER(_vp)]
    lib.sara_hip_sift_group_size.argtypes = [_vp]
    lib.sara_hip_sift_group_context.argtypes = [_vp, C.c_int, C.POINTER(_vp)]
    lib.sara_hip_sift_group_detect.argtypes = [_vp, _vp, _vp, C.c_size_t, C.c_int,
                                               C.c_int, C.c_int, C.c_int, C.c_int]
    lib.sara_hip_sift_group_gather.argtypes = [_vp, C.c_int, C.c_int, _vp,
                                               C.POINTER(_vp), C.POINTER(_vp),
                                               C.POINTER(_vp), C.POINTER(C.c_int)]
    lib.sara_hip_sift_group_destroy.argtypes = [_vp]
    lib.sara_hip_sift_group_collect_host.argtypes = [
        _vp, C.c_int, _vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp),
        C.POINTER(C.c_int)]
    lib.sara_hip_sift_group_transport.argtypes = [_vp]
    lib.sara_hip_sift_group_transport.restype = C.c_char_p
    lib.sara_hip_comm_transport.argtypes = [_vp]
    lib.sara_hip_comm_transport.restype = C.c_char_p
    lib.sara_hip_comm_size.argtypes = [_vp]
    lib.sara_hip_comm_size.restype = C.c_int
    lib.sara_hip_rccl_version.argtypes = [C.POINTER(C.c_int)]
    lib.sara_hip_sift_ticket_counts.argtypes = [_vp, C.c_int, _vp,
                                                C.POINTER(C.c_int),
                                                C.POINTER(C.c_int)]
    lib.sara_hip_sift_collect_into.argtypes = [_vp, C.c_int, _vp, _vp, _vp]
    lib.sara_hip_host_register.argtypes = [_vp, C.c_size_t]
    lib.sara_hip_host_unregister.argtypes = [_vp]
    lib.sara_hip_copy_to_device.argtypes = [_vp, _vp, C.c_size_t, C.c_int]
    lib.sara_hip_device_alloc.argtypes = [C.POINTER(_vp), C.c_size_t, C.c_int]
    lib.sara_hip_device_free.argtypes = [_vp, C.c_int]
    lib.sara_hip_host_alloc.argtypes = [C.POINTER(_vp), C.c_size_t]
    lib.sara_hip_host_free.argtypes = [_vp]
    lib.sara_hip_selfcheck_sincos.argtypes = [_f32p, _f32p, _f32p, C.c_size_t]
    lib.sara_hip_selfcheck_sincos.restype = None
    lib.sara_hip_selfcheck_orientation_bins.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
    lib.sara_hip_selfcheck_definiteness.argtypes = [
        _f32p, C.POINTER(C.c_int), C.c_size_t, C.POINTER(C.c_ubyte), C.c_int]
    return lib


_lib = None


def load():
    """Loads the native library; raises when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "%s is missing: build it with `python -c 'import "
                "__graft_entry__ as g; g.build()'` or `make -C sara_amd/csrc`. "
                "There is no CPU fallback." % LIB_PATH)
        _lib = _declare(C.CDLL(LIB_PATH))
    return _lib


def check(status):
    if status != OK:
        raise SaraHipError(status, load().sara_hip_last_error().decode())


def require_gpu():
    n = load().sara_hip_device_count()
    if n <= 0:
        raise SaraHipError(NO_DEVICE,
                           "no HIP device visible: the MI355X SIFT front-end "
                           "has no CPU fallback")
    return n
