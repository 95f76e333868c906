"""ctypes binding of the CPU oracle (oracle/libsift_ref.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package (sara_amd/) never imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_LIB_PATH = os.path.join(ORACLE_DIR, "libsift_ref.so")

OEREGION_DTYPE = np.dtype(
    {
        "names": ["coords", "shape_matrix", "orientation", "extremum_value",
                  "type", "extremum_type"],
        "formats": [("<f4", 2), ("<f4", 4), "<f4", "<f4", "u1", "i1"],
        "offsets": [0, 16, 32, 36, 40, 41],
        "itemsize": 48,
    }
)

f32p = C.POINTER(C.c_float)
i32p = C.POINTER(C.c_int)


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def _load():
    if not os.path.exists(_LIB_PATH):
        build()
    lib = C.CDLL(_LIB_PATH)
    lib.ref_last_error.restype = C.c_char_p
    lib.ref_sift_run.restype = C.c_void_p
    lib.ref_sift_run.argtypes = [f32p, C.c_int, C.c_int, i32p, f32p, C.c_float,
                                 C.c_float, C.c_float, C.c_int, C.c_int, C.c_int]
    lib.ref_pyramid_run.restype = C.c_void_p
    lib.ref_pyramid_run.argtypes = [f32p, C.c_int, C.c_int, i32p, f32p,
                                    C.c_float, C.c_int]
    lib.ref_sift_free.argtypes = [C.c_void_p]
    for name in ("ref_sift_octave_count", "ref_sift_extrema_count",
                 "ref_sift_keypoint_count"):
        getattr(lib, name).argtypes = [C.c_void_p]
        getattr(lib, name).restype = C.c_int
    lib.ref_sift_octave_info.argtypes = [C.c_void_p, C.c_int, i32p, i32p, f32p]
    for name in ("ref_sift_gaussian", "ref_sift_dog", "ref_sift_gradient"):
        getattr(lib, name).argtypes = [C.c_void_p, C.c_int, C.c_int]
        getattr(lib, name).restype = f32p
    lib.ref_sift_extrema.argtypes = [C.c_void_p, C.c_void_p, i32p]
    lib.ref_sift_keypoints.argtypes = [C.c_void_p, C.c_void_p, i32p, f32p]
    lib.ref_sift_times.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    lib.ref_oeregion_scale.restype = C.c_float
    lib.ref_oeregion_scale.argtypes = [C.c_float]
    lib.ref_rgb8_to_gray32f.restype = C.c_float
    lib.ref_rgb8_to_gray32f.argtypes = [C.c_ubyte, C.c_ubyte, C.c_ubyte]
    return lib


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _load()
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(f32p)


K_DEFAULT = float(np.float32(2.0) ** np.float32(1.0 / 3.0))


class PyramidParams:
    """Mirror of ImagePyramidParams (ImageProcessing/ImagePyramid.hpp:29-52)."""

    def __init__(self, first_octave_index=-1, scale_count_per_octave=6,
                 scale_geometric_factor=None, image_padding_size=1,
                 scale_camera=0.5, scale_initial=1.6,
                 num_octaves_max=2**31 - 1):
        if scale_geometric_factor is None:
            scale_geometric_factor = float(
                np.power(np.float32(2.0), np.float32(1.0) / np.float32(3.0)))
        self.first_octave_index = int(first_octave_index)
        self.scale_count_per_octave = int(scale_count_per_octave)
        self.scale_geometric_factor = float(np.float32(scale_geometric_factor))
        self.image_padding_size = int(image_padding_size)
        self.scale_camera = float(np.float32(scale_camera))
        self.scale_initial = float(np.float32(scale_initial))
        self.num_octaves_max = int(num_octaves_max)

    def ints(self):
        return (C.c_int * 4)(self.first_octave_index,
                             self.scale_count_per_octave,
                             self.image_padding_size, self.num_octaves_max)

    def floats(self):
        return (C.c_float * 3)(self.scale_geometric_factor, self.scale_camera,
                               self.scale_initial)


class RefSift:
    """Result handle of the oracle's compute_sift_keypoints."""

    def __init__(self, image, params, gauss_truncate=4.0, extremum_thres=0.01,
                 edge_ratio_thres=10.0, extremum_refinement_iter=5,
                 parallel=False, stop_after=0, pyramid_only=False):
        img, p = _f(image)
        assert img.ndim == 2
        h, w = img.shape
        if pyramid_only:
            self._h = lib().ref_pyramid_run(p, w, h, params.ints(),
                                            params.floats(), gauss_truncate, 1)
        else:
            self._h = lib().ref_sift_run(p, w, h, params.ints(),
                                         params.floats(), gauss_truncate,
                                         extremum_thres, edge_ratio_thres,
                                         extremum_refinement_iter,
                                         int(parallel), stop_after)
        if not self._h:
            raise RuntimeError(lib().ref_last_error().decode())
        self.params = params

    def __del__(self):
        if getattr(self, "_h", None):
            lib().ref_sift_free(self._h)
            self._h = None

    @property
    def octave_count(self):
        return lib().ref_sift_octave_count(self._h)

    def octave_info(self, o):
        w, h, f = C.c_int(), C.c_int(), C.c_float()
        lib().ref_sift_octave_info(self._h, o, C.byref(w), C.byref(h),
                                   C.byref(f))
        return w.value, h.value, f.value

    def _img(self, fn, s, o, ch=1):
        w, h, _ = self.octave_info(o)
        ptr = fn(self._h, s, o)
        if not ptr:
            return None
        n = w * h * ch
        a = np.ctypeslib.as_array(ptr, shape=(n,)).copy()
        return a.reshape(h, w) if ch == 1 else a.reshape(h, w, ch)

    def gaussian(self, s, o):
        return self._img(lib().ref_sift_gaussian, s, o)

    def dog(self, s, o):
        return self._img(lib().ref_sift_dog, s, o)

    def gradient(self, s, o):
        return self._img(lib().ref_sift_gradient, s, o, 2)

    def extrema(self):
        n = lib().ref_sift_extrema_count(self._h)
        regions = np.zeros(n, dtype=OEREGION_DTYPE)
        xyso = np.zeros((n, 5), dtype=np.int32)
        if n:
            lib().ref_sift_extrema(self._h, regions.ctypes.data,
                                   xyso.ctypes.data_as(i32p))
        return regions, xyso

    def keypoints(self):
        n = lib().ref_sift_keypoint_count(self._h)
        regions = np.zeros(n, dtype=OEREGION_DTYPE)
        so = np.zeros((n, 2), dtype=np.int32)
        desc = np.zeros((n, 128), dtype=np.float32)
        if n:
            lib().ref_sift_keypoints(self._h, regions.ctypes.data,
                                     so.ctypes.data_as(i32p),
                                     desc.ctypes.data_as(f32p))
        return regions, so, desc

    def times(self):
        t = (C.c_double * 7)()
        lib().ref_sift_times(self._h, t)
        names = ["gaussian_pyramid", "dog_pyramid", "dog_extrema", "gradient",
                 "orientation", "descriptors", "total"]
        return dict(zip(names, list(t)))


# ---- unit-level helpers ---------------------------------------------------- #

def convolve_array(signal, kernel, signal_size):
    s, sp = _f(signal)
    s = s.copy()
    sp = s.ctypes.data_as(f32p)
    k, kp = _f(kernel)
    lib().ref_convolve_array(sp, kp, int(signal_size), len(k))
    return s


TAP_VARIANTS = {"expf_serial": 0, "eigen34_sse": 1, "eigen33_sse": 2,
                "ulp_random": 3, "ulp_plus": 4, "ulp_minus": 5,
                "ulp_alternate": 6, "ulp_narrow": 7, "ulp_wide": 8}


def set_tap_variant(kind, seed=0):
    """Arithmetic of make_gaussian_kernel for every later oracle call
    (sift_ref.hpp kTaps*); returns the previous kind.  Process-wide: restore
    it (``with tap_variant(...)``)."""
    kind = TAP_VARIANTS.get(kind, kind)
    lib().ref_set_tap_variant.argtypes = [C.c_int, C.c_uint]
    lib().ref_set_tap_variant.restype = C.c_int
    return lib().ref_set_tap_variant(int(kind), int(seed))


class tap_variant:
    def __init__(self, kind, seed=0):
        self.kind, self.seed = kind, seed

    def __enter__(self):
        self.before = set_tap_variant(self.kind, self.seed)

    def __exit__(self, *exc):
        set_tap_variant(self.before, 0)


def make_gaussian_kernel(sigma, gauss_truncate=4.0):
    out = np.zeros(1024, dtype=np.float32)
    n = lib().ref_make_gaussian_kernel(C.c_float(sigma), C.c_float(gauss_truncate),
                                       out.ctypes.data_as(f32p), 1024)
    assert n > 0
    return out[:n].copy()


def _filter(fn, src, kernel):
    s, sp = _f(src)
    k, kp = _f(kernel)
    h, w = s.shape
    d = np.zeros_like(s)
    fn(sp, d.ctypes.data_as(f32p), w, h, kp, len(k))
    return d


def apply_row_based_filter(src, kernel):
    return _filter(lib().ref_apply_row_based_filter, src, kernel)


def apply_column_based_filter(src, kernel):
    return _filter(lib().ref_apply_column_based_filter, src, kernel)


def apply_gaussian_filter(src, sigma, gauss_truncate=4.0):
    s, sp = _f(src)
    h, w = s.shape
    d = np.zeros_like(s)
    lib().ref_apply_gaussian_filter(sp, d.ctypes.data_as(f32p), w, h,
                                    C.c_float(sigma), C.c_float(gauss_truncate))
    return d


def downscale(src, fact):
    s, sp = _f(src)
    h, w = s.shape
    d = np.zeros((h // fact, w // fact), dtype=np.float32)
    lib().ref_downscale(sp, w, h, int(fact), d.ctypes.data_as(f32p))
    return d


def enlarge(src, dw, dh):
    s, sp = _f(src)
    h, w = s.shape
    d = np.zeros((dh, dw), dtype=np.float32)
    rc = lib().ref_enlarge(sp, w, h, d.ctypes.data_as(f32p), dw, dh)
    if rc:
        raise ValueError(lib().ref_last_error().decode())
    return d


def gradient(src):
    s, sp = _f(src)
    h, w = s.shape
    g = np.zeros((h, w, 2), dtype=np.float32)
    lib().ref_gradient(sp, w, h, g.ctypes.data_as(f32p))
    return g


def hessian(src):
    s, sp = _f(src)
    h, w = s.shape
    g = np.zeros((h, w, 3), dtype=np.float32)
    lib().ref_hessian(sp, w, h, g.ctypes.data_as(f32p))
    return g


def gradient_polar(src):
    s, sp = _f(src)
    h, w = s.shape
    g = np.zeros((h, w, 2), dtype=np.float32)
    lib().ref_gradient_polar(sp, w, h, g.ctypes.data_as(f32p))
    return g


def scale_space_extremum(layers, x, y, strict=False):
    l, lp = _f(layers)
    assert l.shape[0] == 3
    return lib().ref_scale_space_extremum(lp, l.shape[2], l.shape[1], x, y,
                                          int(strict))


def on_edge(img, x, y, edge_ratio):
    s, sp = _f(img)
    return bool(lib().ref_on_edge(sp, s.shape[1], s.shape[0], x, y,
                                  C.c_float(edge_ratio)))


def refine_extremum(layers, x, y, s, type_, val, border_sz, num_iter,
                    scale_initial=1.6, k=K_DEFAULT):
    l, lp = _f(layers)
    pos = (C.c_float * 3)()
    v = C.c_float(val)
    rc = lib().ref_refine_extremum(lp, l.shape[0], l.shape[2], l.shape[1],
                                   C.c_float(scale_initial), C.c_float(k), x, y,
                                   s, type_, pos, C.byref(v), border_sz,
                                   num_iter)
    return rc, np.array(list(pos), dtype=np.float32), np.float32(v.value)


def orientation_histogram(grad, x, y, s, bins=36):
    g, gp = _f(grad)
    h = np.zeros(bins, dtype=np.float32)
    fn = {36: lib().ref_orientation_histogram36,
          24: lib().ref_orientation_histogram24}[bins]
    fn(gp, g.shape[1], g.shape[0], C.c_float(x), C.c_float(y), C.c_float(s),
       h.ctypes.data_as(f32p))
    return h


def lowe_smooth_histogram(hist, iters=6):
    h = np.ascontiguousarray(hist, dtype=np.float32).copy()
    lib().ref_lowe_smooth_histogram36(h.ctypes.data_as(f32p), iters)
    return h


def dominant_orientations(grad, x, y, sigma):
    g, gp = _f(grad)
    out = np.zeros(36, dtype=np.float32)
    hist = np.zeros(36, dtype=np.float32)
    n = lib().ref_dominant_orientations(gp, g.shape[1], g.shape[0],
                                        C.c_float(x), C.c_float(y),
                                        C.c_float(sigma),
                                        out.ctypes.data_as(f32p), 36,
                                        hist.ctypes.data_as(f32p))
    return out[:n].copy(), hist


def sift_descriptor(grad, x, y, s, theta, normalize=True):
    g, gp = _f(grad)
    out = np.zeros(128, dtype=np.float32)
    lib().ref_sift_descriptor(gp, g.shape[1], g.shape[0], C.c_float(x),
                              C.c_float(y), C.c_float(s), C.c_float(theta),
                              int(normalize), out.ctypes.data_as(f32p))
    return out


MATCH_DTYPE = np.dtype([("x_index", "<i4"), ("y_index", "<i4"), ("score", "<f4"),
                        ("rank", "<i4"), ("direction", "<i4")])


def compute_matches(desc1, desc2, sift_ratio_thres):
    """Exhaustive-search restatement of AnnMatcher::compute_matches
    (FeatureMatching/AnnMatcher.cpp:203-268) -> structured array MATCH_DTYPE."""
    a, pa = _f(desc1)
    b, pb = _f(desc2)
    n1, n2 = a.shape[0], b.shape[0]
    dim = a.shape[1] if a.ndim == 2 else b.shape[1]
    fn = lib().ref_compute_matches
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_float,
                   C.c_void_p, C.c_int]
    cap = 2 * (n1 + n2) + 16
    while True:
        out = np.zeros(max(cap, 1), MATCH_DTYPE)
        n = fn(a.ctypes.data, n1, b.ctypes.data, n2, dim, sift_ratio_thres,
               out.ctypes.data, cap)
        if n < 0:
            raise RuntimeError(lib().ref_last_error().decode())
        if n <= cap:
            return out[:n]
        cap = n


def match_features(regions):
    """OERegion records -> the n x 8 float rows the oracle's matcher reads
    (x, y, m00, m10, m01, m11, orientation, type)."""
    r = np.asarray(regions)
    f = np.zeros((len(r), 8), np.float32)
    f[:, 0:2] = r["coords"]
    f[:, 2:6] = r["shape_matrix"]
    f[:, 6] = r["orientation"]
    f[:, 7] = r["type"]
    return f


def compute_self_matches(desc, regions, sift_ratio_thres=1.2,
                         min_max_metric_dist_thres=0.5, pixel_dist_thres=10.0):
    """AnnMatcher{keys, ratio, metric thres, pixel thres}.compute_matches()
    (FeatureMatching/AnnMatcher.cpp:199-268), exhaustive-search restatement."""
    a, _ = _f(desc)
    f = np.ascontiguousarray(match_features(regions))
    n = a.shape[0]
    fn = lib().ref_compute_self_matches
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float,
                   C.c_float, C.c_void_p, C.c_int]
    cap = 4 * n + 16
    while True:
        out = np.zeros(max(cap, 1), MATCH_DTYPE)
        k = fn(a.ctypes.data, f.ctypes.data, n, a.shape[1], sift_ratio_thres,
               min_max_metric_dist_thres, pixel_dist_thres, out.ctypes.data, cap)
        if k < 0:
            raise RuntimeError(lib().ref_last_error().decode())
        if k <= cap:
            return out[:k]
        cap = k


def exhaustive_knn(data, queries, k):
    """The oracle's exhaustive neighbour lists (what it ranks instead of FLANN's
    tree queries): k nearest per query in (distance, index) order."""
    d = np.ascontiguousarray(data, np.float32)
    q = np.ascontiguousarray(queries, np.float32)
    idx = np.zeros((len(q), k), np.int32)
    dist = np.zeros((len(q), k), np.float32)
    fn = lib().ref_exhaustive_knn
    fn.restype = None
    fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int,
                   C.c_void_p, C.c_void_p]
    fn(d.ctypes.data, len(d), d.shape[1], q.ctypes.data, len(q), k,
       idx.ctypes.data, dist.ctypes.data)
    return idx, dist


def exhaustive_radius(data, queries, radii, max_nn=None):
    """Members of the strict radius search per query -> list of (idx, dist)."""
    d = np.ascontiguousarray(data, np.float32)
    q = np.ascontiguousarray(queries, np.float32)
    r = np.ascontiguousarray(radii, np.float32)
    max_nn = max_nn or len(d)
    idx = np.zeros((len(q), max_nn), np.int32)
    dist = np.zeros((len(q), max_nn), np.float32)
    count = np.zeros(len(q), np.int32)
    fn = lib().ref_exhaustive_radius
    fn.restype = None
    fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                   C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    fn(d.ctypes.data, len(d), d.shape[1], q.ctypes.data, len(q), r.ctypes.data,
       max_nn, idx.ctypes.data, dist.ctypes.data, count.ctypes.data)
    return [(idx[i, :count[i]].copy(), dist[i, :count[i]].copy())
            for i in range(len(q))]


def key_proximity(f1, f2, metric_dist_thres=0.5, pixel_dist_thres=10.0):
    """KeyProximity{metric, pixel}(f1, f2), KeyProximity.cpp:17-30; f = 8 floats
    as match_features() lays them out."""
    a = np.ascontiguousarray(f1, np.float32)
    b = np.ascontiguousarray(f2, np.float32)
    fn = lib().ref_key_proximity
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_float]
    return bool(fn(a.ctypes.data, b.ctypes.data, metric_dist_thres,
                   pixel_dist_thres))


def flann_l2(a, b):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    fn = lib().ref_flann_l2
    fn.restype = C.c_float
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    return float(fn(a.ctypes.data, b.ctypes.data, a.size))


MODE_SIGNED_EXTREMUM_TYPE = 1
MODE_DOWNSCALE_AT_DOUBLE_SIGMA = 2


class detector_mode:
    """with detector_mode(bits): ... - the oracle's "corrected" switches."""

    def __init__(self, mode):
        self.mode = int(mode)

    def __enter__(self):
        fn = lib().ref_set_detector_mode
        fn.restype = C.c_int
        fn.argtypes = [C.c_int]
        self.old = fn(self.mode)
        return self

    def __exit__(self, *exc):
        lib().ref_set_detector_mode(self.old)


DEF_EIGEN34, DEF_EIGEN33, DEF_SYLVESTER_DOUBLE = 0, 1, 2


class definiteness_rule:
    """with definiteness_rule(DEF_*): ... - which restatement of the Hessian's
    definiteness test refine_extremum uses (default: Eigen 3.4's float
    SelfAdjointEigenSolver)."""

    def __init__(self, rule):
        self.rule = int(rule)

    def __enter__(self):
        fn = lib().ref_set_definiteness_rule
        fn.restype = C.c_int
        fn.argtypes = [C.c_int]
        self.old = fn(self.rule)
        return self

    def __exit__(self, *exc):
        lib().ref_set_definiteness_rule(self.old)


class definiteness_audit:
    """with definiteness_audit() as a: ...; a.read() -> dict of counters."""

    def __enter__(self):
        lib().ref_definiteness_audit_enable(1)
        return self

    def read(self):
        out = (C.c_longlong * 4)()
        lib().ref_definiteness_audit_read(out)
        return {"sites": out[0], "eigen34_vs_sylvester": out[1],
                "eigen34_vs_eigen33": out[2], "not_converged": out[3]}

    def __exit__(self, *exc):
        self.result = self.read()
        lib().ref_definiteness_audit_enable(0)


class squared_norm_order:
    """with squared_norm_order(1): ... - left-to-right squaredNorm() instead of
    Eigen's Packet4f order in the descriptor's normalize()."""

    def __init__(self, order):
        self.order = int(order)

    def __enter__(self):
        fn = lib().ref_set_squared_norm_order
        fn.restype = C.c_int
        fn.argtypes = [C.c_int]
        self.old = fn(self.order)
        return self

    def __exit__(self, *exc):
        lib().ref_set_squared_norm_order(self.old)


def eigen_squared_norm128(h):
    h = np.ascontiguousarray(h, np.float32)
    assert h.size == 128
    fn = lib().ref_eigen_squared_norm128
    fn.restype = C.c_float
    fn.argtypes = [C.c_void_p]
    return float(fn(h.ctypes.data))


def selfadjoint_eigenvalues3(mats, rule=DEF_EIGEN34):
    m = np.ascontiguousarray(mats, np.float32).reshape(-1, 9)
    lam = np.empty((m.shape[0], 3), np.float32)
    conv = np.empty(m.shape[0], np.int32)
    fn = lib().ref_selfadjoint_eigenvalues3
    fn.restype = None
    fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    fn(m.ctypes.data, m.shape[0], int(rule), lam.ctypes.data, conv.ctypes.data)
    return lam, conv.astype(bool)


def not_definite_enough3(mats, type_, rule=DEF_EIGEN34):
    m = np.ascontiguousarray(mats, np.float32).reshape(-1, 9)
    out = np.empty(m.shape[0], np.int32)
    fn = lib().ref_not_definite_enough3
    fn.restype = None
    fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    fn(m.ctypes.data, m.shape[0], int(type_), int(rule), out.ctypes.data)
    return out.astype(bool)


def halide_dog_extremum_map(a, b, c, edge_ratio=10.0, extremum_thres=0.01):
    """int8 map of the reference's DO_SARA_USE_HALIDE classifier (restated)."""
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    c = np.ascontiguousarray(c, np.float32)
    out = np.zeros(a.shape, np.int8)
    fn = lib().ref_halide_dog_extremum_map
    fn.restype = None
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                   C.c_float, C.c_float, C.c_void_p]
    fn(a.ctypes.data, b.ctypes.data, c.ctypes.data, a.shape[1], a.shape[0],
       float(edge_ratio), float(extremum_thres), out.ctypes.data)
    return out


def root_sift(desc):
    out = np.array(desc, np.float32, order="C", copy=True).reshape(-1, np.shape(desc)[-1])
    fn = lib().ref_root_sift
    fn.restype = None
    fn.argtypes = [C.c_void_p, C.c_int, C.c_int]
    fn(out.ctypes.data, out.shape[0], out.shape[1])
    return out.reshape(np.shape(desc))


def rgb8_to_gray32f(rgb):
    """Vectorised restatement of the reference's Rgb8 -> float conversion
    (double arithmetic, final cast), checked against the C oracle in tests."""
    rgb = np.asarray(rgb, dtype=np.uint8)
    d = rgb.astype(np.float64) / 255.0
    g = 0.2125 * d[..., 0] + 0.7154 * d[..., 1] + 0.0721 * d[..., 2]
    return g.astype(np.float32)
