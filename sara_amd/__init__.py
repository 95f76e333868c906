"""sara_amd — MI355X-native SIFT front-end behind Sara's feature-detection API.

Python-facing mirror of the reference's pybind11 surface
(python/oddkiva/sara/pybind11/FeatureDetectors.cpp:29-125): the same names,
argument meaning and defaults — ``ImagePyramidParams`` (whose *Python* default
first octave is +1, :72-76), ``OERegion``, ``KeypointList``, ``features()``,
``descriptors()``, ``compute_sift_keypoints()`` — over the C-ABI of
include/sara_hip_sift.h.  ``SiftContext`` is the batched, HBM-resident entry
point the benchmark and the multi-GPU path use.

Everything computes on the GPU; there is no CPU path in this package.
"""
import ctypes as C
import threading

import numpy as np

from . import capi
from .capi import (MATCH_DTYPE, OEREGION_DTYPE, STAGE_DESCRIPTOR, STAGE_EXTREMA,
                   STAGE_GRADIENT, STAGE_ORIENTATION, STAGE_PYRAMID,
                   SaraHipError)

__all__ = [
    "ImagePyramidParams", "OERegion", "KeypointList", "features", "descriptors",
    "compute_sift_keypoints", "SiftContext", "ComputeDoGExtrema",
    "apply_gaussian_filter", "gaussian", "scale", "downscale", "enlarge",
    "gradient_polar_coordinates", "scale_space_dog_extremum_map",
    "from_rgb8_to_gray32f", "from_gray8_to_gray32f", "AnnMatcher", "match",
    "MATCH_DTYPE", "match_pairs", "write_keypoints", "read_keypoints", "root_sift", "H5File",
    "make_gaussian_kernel", "SaraHipError", "OEREGION_DTYPE", "DeviceArray",
    "pinned_empty",
]

_K_DEFAULT = float(np.power(np.float32(2.0), np.float32(1.0) / np.float32(3.0)))
_INT_MAX = 2 ** 31 - 1


class ImagePyramidParams:
    """ImageProcessing/ImagePyramid.hpp:29-52 as exposed to Python
    (pybind11/FeatureDetectors.cpp:70-87: default first_octave_index = 1).
    ``num_octaves_max`` is the 7th C++ constructor argument."""

    def __init__(self, first_octave_index=1, scale_count_per_octave=3 + 3,
                 scale_geometric_factor=_K_DEFAULT, image_padding_size=1,
                 scale_camera=0.5, scale_initial=1.6, num_octaves_max=_INT_MAX):
        self._s = capi.PyramidParamsStruct(
            int(first_octave_index), int(scale_count_per_octave),
            float(scale_geometric_factor), int(image_padding_size),
            float(scale_camera), float(scale_initial), int(num_octaves_max))

    first_octave_index = property(lambda s: s._s.first_octave_index)
    scale_count_per_octave = property(lambda s: s._s.scale_count_per_octave)
    scale_geometric_factor = property(lambda s: s._s.scale_geometric_factor)
    image_padding_size = property(lambda s: s._s.image_padding_size)
    scale_camera = property(lambda s: s._s.scale_camera)
    scale_initial = property(lambda s: s._s.scale_initial)
    num_octaves_max = property(lambda s: s._s.num_octaves_max)

    def octave_count(self, width, height):
        """Octaves gaussian_pyramid() builds (GaussianPyramid.hpp:80-94)."""
        return capi.load().sara_hip_pyramid_octave_count(C.byref(self._s),
                                                         int(width), int(height))

    def octave_info(self, width, height, octave):
        w, h, f = C.c_int(), C.c_int(), C.c_float()
        capi.check(capi.load().sara_hip_pyramid_octave_info(
            C.byref(self._s), int(width), int(height), int(octave), C.byref(w),
            C.byref(h), C.byref(f)))
        return w.value, h.value, f.value


class OERegion:
    """Features/Feature.hpp:40-179 (fields of the pybind class :31-39)."""

    __slots__ = ("coords", "shape_matrix", "orientation", "extremum_value",
                 "type", "extremum_type")

    def __init__(self, rec=None):
        if rec is None:
            self.coords = np.zeros(2, np.float32)
            self.shape_matrix = np.zeros((2, 2), np.float32)
            self.orientation = 0.0
            self.extremum_value = 0.0
            self.type = 11
            self.extremum_type = -2
        else:
            self.coords = np.array(rec["coords"], np.float32)
            # column-major 2x2
            self.shape_matrix = np.array(rec["shape_matrix"],
                                         np.float32).reshape(2, 2).T
            self.orientation = float(rec["orientation"])
            self.extremum_value = float(rec["extremum_value"])
            self.type = int(rec["type"])
            self.extremum_type = int(rec["extremum_type"])

    def radius(self, radian=0.0):
        """Features/Feature.cpp:28-39."""
        u, s, _ = np.linalg.svd(self.shape_matrix.astype(np.float32))
        radii = 1.0 / np.sqrt(s)
        d = np.array([np.cos(radian), np.sin(radian)], np.float32)
        x = radii[0] * u[:, 0].dot(d)
        y = radii[1] * u[:, 1].dot(d)
        return float(np.sqrt(x * x + y * y))

    def __eq__(self, other):
        return (np.array_equal(self.coords, other.coords)
                and np.array_equal(self.shape_matrix, other.shape_matrix)
                and self.orientation == other.orientation
                and self.type == other.type)


class KeypointList:
    """Features/KeypointList.hpp:35-96: (features, N x 128 descriptors), plus
    the (s, o) pairs the detector reports (DoG.cpp:75-81)."""

    def __init__(self, regions=None, descriptor_matrix=None, scale_octave=None):
        self.regions = (np.zeros(0, OEREGION_DTYPE) if regions is None
                        else regions)
        self.descriptor_matrix = (np.zeros((0, 128), np.float32)
                                  if descriptor_matrix is None
                                  else descriptor_matrix)
        self.scale_octave = (np.zeros((0, 2), np.int32) if scale_octave is None
                             else scale_octave)

    def __len__(self):
        return len(self.regions)


def features(keys):
    """List of OERegion (pybind11/FeatureDetectors.cpp:57-62)."""
    return [OERegion(r) for r in keys.regions]


def descriptors(keys):
    """N x 128 float32 matrix (pybind11/FeatureDetectors.cpp:63-68)."""
    return keys.descriptor_matrix


#: Options (capi.OPT_* -> value) every NEW SiftContext gets right after it is
#: created - also the contexts the free functions and ComputeDoGExtrema build
#: and cache (the cache key includes them).  E.g. the parity tests run the
#: real-image pack under {capi.OPT_KERNEL_SELECTION: capi.SELECT_SHIPPED} and
#: ..._FORCED_MARCH in one process.  ``with sara_amd.default_options({...})``.
DEFAULT_OPTIONS = {}


class default_options:
    def __init__(self, options):
        self.options = dict(options)

    def __enter__(self):
        self.before = dict(DEFAULT_OPTIONS)
        DEFAULT_OPTIONS.update(self.options)
        return self

    def __exit__(self, *exc):
        DEFAULT_OPTIONS.clear()
        DEFAULT_OPTIONS.update(self.before)


class SiftContext:
    """Batched compute_sift_keypoints with all stages resident in HBM.

    ``dog_args=(img_padding_sz, extremum_refinement_iter)`` builds the context
    with ComputeDoGExtrema's own constructor arguments (DoG.hpp:72-78) instead
    of compute_sift_keypoints' shifted ones (SIFT.cpp:45-51).
    """

    def __init__(self, max_width, max_height, max_batch=1,
                 pyramid_params=None, gauss_truncate=4.0, extremum_thres=0.01,
                 edge_ratio_thres=10.0, extremum_refinement_iter=5,
                 max_keypoints=0, device=0, dog_args=None):
        lib = capi.load()
        capi.require_gpu()
        self.params = pyramid_params or ImagePyramidParams()
        self._h = C.c_void_p()
        if dog_args is None:
            sp = capi.SiftParamsStruct(self.params._s, gauss_truncate,
                                       extremum_thres, edge_ratio_thres,
                                       int(extremum_refinement_iter))
            st = lib.sara_hip_sift_create(C.byref(sp), max_width, max_height,
                                          max_batch, max_keypoints, device,
                                          C.byref(self._h))
        else:
            st = lib.sara_hip_sift_create_dog(
                C.byref(self.params._s), gauss_truncate, extremum_thres,
                edge_ratio_thres, int(dog_args[0]), int(dog_args[1]), max_width,
                max_height, max_batch, max_keypoints, device, C.byref(self._h))
        capi.check(st)
        self.device = device
        self.max_width, self.max_height = max_width, max_height
        self.max_batch = max_batch
        self.batch = 0
        self._keepalive = None
        for option, value in sorted(DEFAULT_OPTIONS.items()):
            self.set_option(option, value)

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            try:
                capi.load().sara_hip_sift_destroy(self._h)
            except (AttributeError, TypeError):
                pass  # interpreter shutdown: the modules are already gone
            self._h = C.c_void_p()

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_option(self, option, value):
        capi.check(capi.load().sara_hip_sift_set_option(self._h, option, value))

    # -- detection ---------------------------------------------------------- #
    def detect(self, images, last_stage=STAGE_DESCRIPTOR, stream=None):
        """images: H x W or B x H x W float32 (host numpy array)."""
        a = np.ascontiguousarray(images, dtype=np.float32)
        if a.ndim == 2:
            a = a[None]
        if a.ndim != 3:
            raise ValueError("images must be H x W or B x H x W")
        b, h, w = a.shape
        self._keepalive = a
        self.batch = b
        capi.check(capi.load().sara_hip_sift_detect(
            self._h, a.ctypes.data, 0, b, w, h, 0, int(last_stage), stream))
        return self

    def detect_u8(self, images, last_stage=STAGE_DESCRIPTOR, stream=None):
        """8-bit frames converted on the device: H x W or B x H x W (gray8),
        H x W x 3 or B x H x W x 3 (RGB8, from_rgb8_to_gray32f semantics)."""
        a = np.ascontiguousarray(images, dtype=np.uint8)
        if a.ndim == 2 or (a.ndim == 3 and a.shape[-1] == 3):
            a = a[None]
        if a.ndim == 3:
            channels = 1
        elif a.ndim == 4 and a.shape[-1] == 3:
            channels = 3
        else:
            raise ValueError("images must be (B,)H x W or (B,)H x W x 3 uint8")
        b, h, w = a.shape[:3]
        self._keepalive = a
        self.batch = b
        capi.check(capi.load().sara_hip_sift_detect_u8(
            self._h, a.ctypes.data, 0, channels, b, w, h, 0, int(last_stage),
            stream))
        return self

    def stage(self, images):
        """Start the upload of the NEXT batch (float32 B x H x W, uint8 B x H x
        W, or uint8 B x H x W x 3) on the copy stream; returns at once.  The
        array is kept alive until the next stage()."""
        a = np.ascontiguousarray(images)
        if a.dtype == np.uint8:
            if a.ndim == 2 or (a.ndim == 3 and a.shape[-1] == 3):
                a = a[None]
            channels = 3 if a.ndim == 4 else 1
        else:
            a = np.ascontiguousarray(a, dtype=np.float32)
            if a.ndim == 2:
                a = a[None]
            channels = 0
        b, h, w = a.shape[:3]
        self._staged_keepalive = a
        self._staged_batch = b
        capi.check(capi.load().sara_hip_sift_stage(
            self._h, a.ctypes.data, 0, channels, b, w, h))
        return self

    def detect_staged(self, last_stage=STAGE_DESCRIPTOR, stream=None):
        """Run the pipeline on the batch staged last."""
        self.batch = self._staged_batch
        self._keepalive = self._staged_keepalive
        capi.check(capi.load().sara_hip_sift_detect_staged(
            self._h, int(last_stage), stream))
        return self

    def detect_device(self, ptr, batch, width, height, frame_stride=0,
                      last_stage=STAGE_DESCRIPTOR, stream=None):
        """Frames already resident in HBM at raw device pointer ``ptr``."""
        self.batch = batch
        capi.check(capi.load().sara_hip_sift_detect(
            self._h, ptr, frame_stride, batch, width, height, 1,
            int(last_stage), stream))
        return self

    def synchronize(self):
        capi.check(capi.load().sara_hip_sift_synchronize(self._h))

    # -- keypoint-list capacity ------------------------------------------------ #
    def capacity(self):
        """-> (max_keypoints per frame, capacity the last examined batch asked
        for); sara_hip_sift_capacity."""
        cap, need = C.c_int(0), C.c_int(0)
        capi.check(capi.load().sara_hip_sift_capacity(self._h, C.byref(cap),
                                                      C.byref(need)))
        return cap.value, need.value

    def reserve(self, max_keypoints):
        """Grow the per-frame list capacity (sara_hip_sift_reserve)."""
        capi.check(capi.load().sara_hip_sift_reserve(self._h, int(max_keypoints)))
        return self

    def grow_for_last_batch(self):
        """After a SARA_HIP_CAPACITY_EXCEEDED: reserve twice what the batch
        asked for (a list that overflowed starves the ones behind it, so the
        figure is a lower bound and the caller loops)."""
        cap, need = self.capacity()
        return self.reserve(2 * max(cap, need))

    # -- pipelined host-to-host operation ------------------------------------- #
    def submit(self, images, last_stage=STAGE_DESCRIPTOR):
        """Enqueue one batch (float32 B x H x W, uint8 B x H x W, or uint8 B x H
        x W x 3, host arrays) and return its ticket at once.  Up to two batches
        may be in flight: call submit(i + 1) before collect(i) to overlap the
        upload and kernels of one batch with the read-back of the other."""
        a = np.ascontiguousarray(images)
        if a.dtype == np.uint8:
            if a.ndim == 2 or (a.ndim == 3 and a.shape[-1] == 3):
                a = a[None]
            channels = 3 if a.ndim == 4 else 1
        else:
            a = np.ascontiguousarray(a, dtype=np.float32)
            if a.ndim == 2:
                a = a[None]
            channels = 0
        b, h, w = a.shape[:3]
        ticket = C.c_int(-1)
        capi.check(capi.load().sara_hip_sift_submit(
            self._h, a.ctypes.data, 0, channels, b, w, h, 0, int(last_stage),
            C.byref(ticket)))
        self._inflight = getattr(self, "_inflight", {})
        self._inflight[ticket.value] = (a, b)
        return ticket.value

    def stage_raw(self, ptr, channels, batch, width, height, frame_stride=0):
        """stage() from a raw host pointer (the caller keeps the memory alive
        and unmodified until the batch has been collected)."""
        self._staged_keepalive = None
        self._staged_batch = batch
        capi.check(capi.load().sara_hip_sift_stage(
            self._h, ptr, frame_stride, channels, batch, width, height))
        return self

    def submit_staged(self, last_stage=STAGE_DESCRIPTOR):
        """submit() for the batch stage() put on its way: ``stage(i + 1);
        collect(i - 1); submit_staged(i + 1)`` starts the next upload before
        the host waits for a read-back (sara_hip_sift_submit_staged)."""
        ticket = C.c_int(-1)
        capi.check(capi.load().sara_hip_sift_submit_staged(
            self._h, int(last_stage), C.byref(ticket)))
        self._inflight = getattr(self, "_inflight", {})
        self._inflight[ticket.value] = (self._staged_keepalive, self._staged_batch)
        return ticket.value

    def submit_raw(self, ptr, channels, batch, width, height, on_device=False,
                   frame_stride=0, last_stage=STAGE_DESCRIPTOR):
        """submit() from a raw pointer: pinned / pageable host memory, or HBM
        when ``on_device``.  channels: 0 float32, 1 gray8, 3 RGB8.  The caller
        keeps the memory alive until the ticket is collected."""
        ticket = C.c_int(-1)
        capi.check(capi.load().sara_hip_sift_submit(
            self._h, ptr, frame_stride, channels, batch, width, height,
            1 if on_device else 0, int(last_stage), C.byref(ticket)))
        self._inflight = getattr(self, "_inflight", {})
        self._inflight[ticket.value] = (None, batch)
        return ticket.value

    def collect(self, ticket, with_descriptors=True, copy=False):
        """-> (frame_offsets[B+1], regions[N], descriptors[N,128] or None,
        scale_octave[N,2]) of the batch of ``ticket``: views of pinned host
        memory owned by the context (valid until the second submit() after
        this ticket's) unless ``copy``."""
        _, batch = getattr(self, "_inflight", {}).pop(ticket, (None, 0))
        f, d, s, o = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        total = C.c_int(0)
        st = capi.load().sara_hip_sift_collect(
            self._h, ticket, C.byref(f),
            C.byref(d) if with_descriptors else None, C.byref(s), C.byref(o),
            C.byref(total))
        capi.check(st)
        n = total.value

        def view(ptr, count, dtype, shape):
            if count == 0 or not ptr.value:
                return np.zeros(shape, dtype)
            nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
            buf = (C.c_char * nbytes).from_address(ptr.value)
            raw = np.frombuffer(buf, dtype=np.uint8)
            if copy:
                # byte-wise: a field-wise copy of the padded OERegion dtype
                # would leave the padding bytes undefined
                raw = raw.copy()
            return raw.view(dtype).reshape(shape)

        offsets = view(o, batch + 1, np.int32, (batch + 1,))
        regions = view(f, n, OEREGION_DTYPE, (n,))
        so = view(s, n, np.int32, (n, 2))
        desc = view(d, n, np.float32, (n, 128)) if with_descriptors else None
        return offsets, regions, desc, so

    def ticket_counts(self, ticket):
        """Waits for the batch of ``ticket``; -> (frame_offsets[B+1], total).
        The ticket stays pending."""
        off = np.zeros(self.max_batch + 1, np.int32)
        b, total = C.c_int(0), C.c_int(0)
        capi.check(capi.load().sara_hip_sift_ticket_counts(
            self._h, ticket, off.ctypes.data, C.byref(b), C.byref(total)))
        return off[:b.value + 1], total.value

    def collect_into(self, ticket, features_ptr, descriptors_ptr, scale_octave_ptr):
        """Read the batch of ``ticket`` back into caller-owned host memory (raw
        addresses, 0 / None to skip an array) and consume the ticket: several
        ranks deliver into ONE shared host array at their global offsets."""
        getattr(self, "_inflight", {}).pop(ticket, None)
        capi.check(capi.load().sara_hip_sift_collect_into(
            self._h, ticket, features_ptr or None, descriptors_ptr or None,
            scale_octave_ptr or None))

    def counts(self):
        c = np.zeros(self.batch, np.int32)
        tot = C.c_int32()
        capi.check(capi.load().sara_hip_sift_counts(
            self._h, c.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(tot)))
        return c, tot.value

    def fetch(self, with_descriptors=True):
        """-> (per-frame counts, regions[N], descriptors[N,128] or None,
        scale_octave[N,2]) for the whole batch, frames concatenated."""
        c, total = self.counts()
        regions = np.zeros(total, OEREGION_DTYPE)
        so = np.zeros((total, 2), np.int32)
        desc = np.zeros((total, 128), np.float32) if with_descriptors else None
        capi.check(capi.load().sara_hip_sift_fetch(
            self._h, regions.ctypes.data,
            desc.ctypes.data if with_descriptors else None, so.ctypes.data, 0))
        return c, regions, desc, so

    def device_results(self):
        f, d, s, o = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        capi.check(capi.load().sara_hip_sift_device_results(
            self._h, C.byref(f), C.byref(d), C.byref(s), C.byref(o)))
        return f.value, d.value, s.value, o.value

    def match_frames(self, i, j, lowe_ratio):
        """match() (SfM/Helpers/KeypointMatching.cpp:19-25) between the
        keypoints of frames i and j of the last detect(), descriptors read
        where they are in HBM (sara_hip_sift_device_results)."""
        counts, _ = self.counts()
        off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        _, d_desc, _, _ = self.device_results()
        n1, n2 = int(counts[i]), int(counts[j])
        cap = 2 * (n1 + n2) + 16
        for _ in range(2):
            out = np.zeros(max(cap, 1), MATCH_DTYPE)
            count = C.c_int(0)
            st = capi.load().sara_hip_match_descriptors(
                d_desc + int(off[i]) * 512, n1, d_desc + int(off[j]) * 512, n2,
                128, float(lowe_ratio), 1, out.ctypes.data, cap, C.byref(count),
                self.device)
            if st == capi.CAPACITY_EXCEEDED and count.value > cap:
                cap = count.value
                continue
            capi.check(st)
            break
        return out[:count.value]

    def match_frame_pairs(self, frame_pairs, lowe_ratio):
        """match_frames() for several (i, j) pairs of the last detect() in one
        call (sara_hip_match_descriptors_batch): e.g. consecutive frames of a
        batch, [(0, 1), (1, 2), ...].  Returns one match array per pair."""
        counts, _ = self.counts()
        off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        _, d_desc, _, _ = self.device_results()
        pairs = [(d_desc + int(off[i]) * 512, int(counts[i]),
                  d_desc + int(off[j]) * 512, int(counts[j]))
                 for i, j in frame_pairs]
        if not pairs:
            return []
        cap = sum(2 * (n1 + n2) + 16 for _, n1, _, n2 in pairs)
        return _match_batch(pairs, 128, lowe_ratio, 1, self.device, cap)

    def keypoint_lists(self, with_descriptors=True):
        c, regions, desc, so = self.fetch(with_descriptors)
        out, at = [], 0
        for n in c:
            n = int(n)
            out.append(KeypointList(regions[at:at + n],
                                    desc[at:at + n] if desc is not None else None,
                                    so[at:at + n]))
            at += n
        return out

    # -- ComputeDoGExtrema accessors ------------------------------------------ #
    @property
    def octave_count(self):
        return capi.load().sara_hip_sift_octave_count(self._h)

    def octave_info(self, o):
        w, h, f = C.c_int(), C.c_int(), C.c_float()
        capi.check(capi.load().sara_hip_sift_octave_info(
            self._h, o, C.byref(w), C.byref(h), C.byref(f)))
        return w.value, h.value, f.value

    def _plane(self, fn, frame, s, o, ch):
        w, h, _ = self.octave_info(o)
        out = np.zeros((h, w) if ch == 1 else (h, w, ch), np.float32)
        capi.check(fn(self._h, frame, s, o,
                      out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def gaussian(self, s, o, frame=0):
        return self._plane(capi.load().sara_hip_sift_copy_gaussian, frame, s, o, 1)

    def dog(self, s, o, frame=0):
        return self._plane(capi.load().sara_hip_sift_copy_dog, frame, s, o, 1)

    def gradient(self, s, o, frame=0):
        return self._plane(capi.load().sara_hip_sift_copy_gradient, frame, s, o, 2)

    def extrema(self):
        """-> (per-frame counts, regions[N], xyso_type[N,5]) in (o, s, raster)
        order, before orientation assignment."""
        lib = capi.load()
        c = np.zeros(self.batch, np.int32)
        tot = C.c_int32()
        capi.check(lib.sara_hip_sift_extrema_counts(
            self._h, c.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(tot)))
        regions = np.zeros(tot.value, OEREGION_DTYPE)
        xyso = np.zeros((tot.value, 5), np.int32)
        capi.check(lib.sara_hip_sift_fetch_extrema(
            self._h, regions.ctypes.data,
            xyso.ctypes.data_as(C.POINTER(C.c_int32))))
        return c, regions, xyso

    def pyramid_launches(self):
        """Per-launch device times of the pyramid stage of the last detect()
        (capi.OPT_LAUNCH_TIMERS; with capi.OPT_SINGLE_STREAM they are kernel
        durations) -> structured array (octave, scale, taps, pixels, ms)."""
        out = np.zeros(256, capi.LAUNCH_TIME_DTYPE)
        n = C.c_int(0)
        capi.check(capi.load().sara_hip_sift_pyramid_launches(
            self._h, out.ctypes.data, len(out), C.byref(n)))
        return out[:min(n.value, len(out))]

    def stage_times(self):
        ms = (C.c_float * 7)()
        capi.check(capi.load().sara_hip_sift_stage_times(self._h, ms))
        return dict(zip(capi.TIME_NAMES, list(ms)))


class DeviceArray:
    """A host array uploaded once into HBM (sara_hip_device_alloc +
    sara_hip_copy_to_device): ``ptr`` goes to detect_device() / submit_raw(...,
    on_device=True).  Freed by close() / the context manager / the collector."""

    def __init__(self, array, device=0):
        a = np.ascontiguousarray(array)
        self.shape, self.dtype, self.device = a.shape, a.dtype, device
        p = C.c_void_p()
        capi.check(capi.load().sara_hip_device_alloc(C.byref(p), max(a.nbytes, 1),
                                                     device))
        self.ptr = p.value
        if a.nbytes:
            capi.check(capi.load().sara_hip_copy_to_device(self.ptr, a.ctypes.data,
                                                           a.nbytes, device))

    def close(self):
        if getattr(self, "ptr", None):
            capi.load().sara_hip_device_free(self.ptr, self.device)
            self.ptr = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown
            pass


def pinned_empty(shape, dtype=np.float32):
    """numpy array in pinned host memory (sara_hip_host_alloc = hipHostMalloc):
    what submit() / stage() upload fastest from.  The memory lives as long as
    the array (and its views)."""
    dt = np.dtype(dtype)
    n = int(np.prod(shape)) * dt.itemsize
    ptr = C.c_void_p()
    capi.check(capi.load().sara_hip_host_alloc(C.byref(ptr), max(n, 1)))

    class _Owner:
        def __init__(self, p):
            self.p = p

        def __del__(self):
            try:
                capi.load().sara_hip_host_free(self.p)
            except Exception:  # interpreter shutdown
                pass

    buf = (C.c_char * max(n, 1)).from_address(ptr.value)
    buf._owner = _Owner(ptr)
    arr = np.frombuffer(buf, dtype=dt, count=int(np.prod(shape))).reshape(shape)
    return arr


def compute_sift_keypoints(image, pyramid_params=None, gauss_truncate=4.0,
                           extremum_thres=0.01, edge_ratio_thres=10.0,
                           extremum_refinement_iter=5, parallel=True, device=0):
    """FeatureDetectors/SIFT.hpp:24-33 / pybind11/FeatureDetectors.cpp:116-124.

    ``image``: H x W float32.  ``parallel`` is accepted for signature
    compatibility (the GPU path is always parallel)."""
    del parallel
    img = np.ascontiguousarray(image, dtype=np.float32)
    if img.ndim != 2:
        raise ValueError("image must be a 2-D float32 array")
    h, w = img.shape
    params = pyramid_params or ImagePyramidParams()
    ctx = _cached_context(w, h, params, gauss_truncate, extremum_thres,
                          edge_ratio_thres, extremum_refinement_iter, device)
    # The reference's lists are std::vectors (RefineExtremum.cpp:496-514): an
    # image never has "too many" keypoints.  The context's lists are sized for
    # ordinary frames; one that overflows them is run again on grown lists, and
    # the grown context stays cached (a video pays once).
    for _ in range(_MAX_GROWTH_STEPS):
        ticket = ctx.submit(img)
        try:
            _, regions, desc, so = ctx.collect(ticket, copy=True)
        except capi.SaraHipError as e:
            if e.status != capi.CAPACITY_EXCEEDED:
                raise
            ctx.grow_for_last_batch()
            continue
        return KeypointList(regions, desc, so)
    raise capi.SaraHipError(capi.CAPACITY_EXCEEDED,
                            "keypoint lists still overflow after %d growth steps"
                            % _MAX_GROWTH_STEPS)


#: contexts of compute_sift_keypoints(), most recently used first, ONE LIST PER
#: THREAD (like the thread_local cache of include/DO/Sara/HipSift.hpp): a
#: context is not thread-safe, and a shared list would let one thread evict -
#: and destroy - a context another thread is still inside submit()/collect()
#: with.  Creating a context allocates the pyramid / gradient / list buffers
#: in HBM (tens of milliseconds); a detection takes less than one.
_CONTEXT_CACHE_MAX = 4
#: every step at least doubles the lists: 16 200 -> over 4 M entries per frame
_MAX_GROWTH_STEPS = 8


class _ThreadContexts(threading.local):
    def __init__(self):
        # dropped with the thread; SiftContext.__del__ then frees its HBM
        self.entries = []


_CONTEXTS = _ThreadContexts()


def _cached_context(w, h, params, gauss_truncate, extremum_thres,
                    edge_ratio_thres, extremum_refinement_iter, device):
    s = params._s
    key = (w, h, device, s.first_octave_index,
           s.scale_count_per_octave, s.scale_geometric_factor,
           s.image_padding_size, s.scale_camera, s.scale_initial,
           s.num_octaves_max, float(gauss_truncate), float(extremum_thres),
           float(edge_ratio_thres), int(extremum_refinement_iter),
           tuple(sorted(DEFAULT_OPTIONS.items())))
    cache = _CONTEXTS.entries
    for i, (k, ctx) in enumerate(cache):
        if k == key:
            if i:
                cache.insert(0, cache.pop(i))
            return ctx
    ctx = SiftContext(w, h, 1, params, gauss_truncate, extremum_thres,
                      edge_ratio_thres, extremum_refinement_iter, device=device)
    cache.insert(0, (key, ctx))
    while len(cache) > _CONTEXT_CACHE_MAX:
        cache.pop()[1].close()      # only ever a context of THIS thread
    return ctx


def clear_context_cache():
    """Release the contexts compute_sift_keypoints() keeps for the calling
    thread."""
    cache = _CONTEXTS.entries
    while cache:
        cache.pop()[1].close()


class ComputeDoGExtrema:
    """FeatureDetectors/DoG.hpp:72-165: functor keeping the pyramids."""

    def __init__(self, pyramid_params=None, gauss_truncate=4.0,
                 extremum_thres=0.01, edge_ratio_thres=10.0, img_padding_sz=1,
                 extremum_refinement_iter=5, device=0):
        self.params = pyramid_params or ImagePyramidParams()
        if self.params.scale_count_per_octave < 4:
            raise RuntimeError("Error: The extraction of DoG extrema needs "
                               "(1 + 3) = 4 scales per octave at the very "
                               "minimum!")
        self._args = (gauss_truncate, extremum_thres, edge_ratio_thres,
                      img_padding_sz, extremum_refinement_iter, device)
        self._ctx = None
        self._ctx_key = None

    def __call__(self, image):
        """-> (regions, scale_octave pairs [N,2])."""
        img = np.ascontiguousarray(image, dtype=np.float32)
        h, w = img.shape
        gt, et, er, pad, it, dev = self._args
        # the reference builds its pyramids from the functor's members on every
        # call (DoG.cpp:23-45): a params member changed between two calls counts
        s = self.params._s
        key = (w, h, s.first_octave_index, s.scale_count_per_octave,
               s.scale_geometric_factor, s.image_padding_size, s.scale_camera,
               s.scale_initial, s.num_octaves_max,
               tuple(sorted(DEFAULT_OPTIONS.items())))
        if self._ctx is None or self._ctx_key != key:
            if self._ctx is not None:
                self._ctx.close()
            self._ctx = SiftContext(w, h, 1, self.params, gt, et, er,
                                    dog_args=(pad, it), device=dev)
            self._ctx_key = key
        # no capacity in the reference (RefineExtremum.cpp:496-514): grow and
        # run again when a frame overflows the lists
        for _ in range(_MAX_GROWTH_STEPS):
            self._ctx.detect(img, last_stage=STAGE_EXTREMA)
            try:
                _, regions, xyso = self._ctx.extrema()
            except capi.SaraHipError as e:
                if e.status != capi.CAPACITY_EXCEEDED:
                    raise
                self._ctx.grow_for_last_batch()
                continue
            return regions, xyso[:, 2:4].copy()
        raise capi.SaraHipError(capi.CAPACITY_EXCEEDED,
                                "extremum lists still overflow after %d growth "
                                "steps" % _MAX_GROWTH_STEPS)

    def gaussians(self, s, o):
        return self._ctx.gaussian(s, o)

    def diff_of_gaussians(self, s, o):
        return self._ctx.dog(s, o)


# ---- operator-level seams --------------------------------------------------- #

def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


def make_gaussian_kernel(sigma, gauss_truncate=4.0,
                         arithmetic=capi.TAPS_LIBM_SERIAL):
    """LinearFiltering.hpp:171-203 (host arithmetic, no GPU needed).
    ``arithmetic``: capi.TAPS_* - how exp() and sum() are evaluated (a scalar
    build of the reference, or the Eigen 3.4 / 3.3 SSE2 packet paths)."""
    out = np.zeros(1024, np.float32)
    n = capi.load().sara_hip_make_gaussian_kernel_with(
        arithmetic, sigma, gauss_truncate,
        out.ctypes.data_as(C.POINTER(C.c_float)), 1024)
    if n <= 0:
        raise ValueError("kernel too large")
    return out[:n].copy()


def apply_gaussian_filter(src, sigma, gauss_truncate=4.0, device=0):
    """LinearFiltering.cpp:30-68."""
    s, sp = _f32(src)
    d = np.zeros_like(s)
    capi.check(capi.load().sara_hip_apply_gaussian_filter(
        sp, d.ctypes.data_as(C.POINTER(C.c_float)), s.shape[1], s.shape[0],
        sigma, gauss_truncate, device))
    return d


gaussian = apply_gaussian_filter


def scale(src, dst_width, dst_height, device=0):
    """Resize.cpp:31-62 (nearest neighbour)."""
    s, sp = _f32(src)
    d = np.zeros((dst_height, dst_width), np.float32)
    capi.check(capi.load().sara_hip_scale(
        sp, s.shape[1], s.shape[0], d.ctypes.data_as(C.POINTER(C.c_float)),
        dst_width, dst_height, device))
    return d


def downscale(src, fact, device=0):
    """Resize.cpp:64-84."""
    h, w = np.asarray(src).shape
    return scale(src, w // fact, h // fact, device)


def enlarge(src, dst_width, dst_height, device=0):
    """Resize.cpp:86-128 (bilinear)."""
    s, sp = _f32(src)
    d = np.zeros((dst_height, dst_width), np.float32)
    capi.check(capi.load().sara_hip_enlarge(
        sp, s.shape[1], s.shape[0], d.ctypes.data_as(C.POINTER(C.c_float)),
        dst_width, dst_height, device))
    return d


def from_rgb8_to_gray32f(rgb, device=0):
    """ImageProcessing/FastColorConversion.cpp:42-66: H x W x 3 uint8 ->
    H x W float32."""
    a = np.ascontiguousarray(rgb, dtype=np.uint8)
    if a.ndim != 3 or a.shape[2] != 3:
        raise ValueError("rgb must be H x W x 3 uint8")
    out = np.zeros(a.shape[:2], np.float32)
    capi.check(capi.load().sara_hip_from_rgb8_to_gray32f(
        a.ctypes.data, out.ctypes.data_as(C.POINTER(C.c_float)), a.shape[1],
        a.shape[0], device))
    return out


def from_gray8_to_gray32f(src, device=0):
    a = np.ascontiguousarray(src, dtype=np.uint8)
    out = np.zeros(a.shape, np.float32)
    capi.check(capi.load().sara_hip_from_gray8_to_gray32f(
        a.ctypes.data, out.ctypes.data_as(C.POINTER(C.c_float)), a.shape[1],
        a.shape[0], device))
    return out


def root_sift(desc, device=0):
    """FeatureDescriptors/RootSIFT.hpp:45-53 as a post-processing of a
    descriptor matrix: rows divided by their L1 norm, then the square root of
    every bin.  Returns a new N x dim float32 array (computed on the GPU)."""
    out = np.array(desc, dtype=np.float32, order="C", copy=True)
    if out.ndim != 2 or out.shape[1] < 1:
        raise ValueError("desc must be N x dim")
    capi.check(capi.load().sara_hip_root_sift(out.ctypes.data, out.shape[0],
                                              out.shape[1], 0, device))
    return out


class AnnMatcher:
    """FeatureMatching/AnnMatcher.hpp:32-86, both constructors:

    ``AnnMatcher(keys1, keys2, sift_ratio_thres=1.2)`` - two key sets;
    ``AnnMatcher(keys, sift_ratio_thres=1.2, min_max_metric_dist_thres=0.5,
    pixel_dist_thres=10.0)`` - one key set against itself, neighbours that
    KeyProximity finds too close dropped (needs a KeypointList: the filter
    reads the OERegions).

    Lowe's ratio on squared L2 distances in both directions, duplicates
    removed, sorted by score; a ratio above 1 (the default) is the reference's
    adaptive radius search and yields matches of rank > 1.  The neighbour
    searches are exact on the GPU (the reference's FLANN kd-trees approximate
    them).  ``compute_matches`` returns a structured array (x_index, y_index,
    score, rank, direction)."""

    def __init__(self, keys1, keys2=None, *args, device=0, **kw):
        names = ("sift_ratio_thres", "min_max_metric_dist_thres", "pixel_dist_thres")
        self._self_matching = keys2 is None or np.isscalar(keys2)
        if self._self_matching:
            # AnnMatcher(keys, ratio, metric_thres, pixel_thres)
            pos = ([] if keys2 is None else [keys2]) + list(args)
            opt = dict(zip(names, pos))
            opt.update(kw)
            if not isinstance(keys1, KeypointList):
                raise TypeError("self-matching needs a KeypointList (features "
                                "and descriptors)")
            self._d1 = self._d2 = self._descriptors(keys1)
            self._regions = np.ascontiguousarray(keys1.regions)
            if len(self._regions) != len(self._d1):
                raise RuntimeError("The list of keypoints are inconsistent in size!")
            self._metric = float(opt.get("min_max_metric_dist_thres", 0.5))
            self._pixel = float(opt.get("pixel_dist_thres", 10.0))
        else:
            opt = dict(zip(names[:1], args))
            opt.update(kw)
            self._d1 = self._descriptors(keys1)
            self._d2 = self._descriptors(keys2)
            if (isinstance(keys1, KeypointList) and
                    len(keys1.regions) != len(self._d1)) or \
               (isinstance(keys2, KeypointList) and
                    len(keys2.regions) != len(self._d2)):
                # AnnMatcher.cpp:181-184
                raise RuntimeError("The list of keypoints are inconsistent in size!")
        self._ratio = float(opt.get("sift_ratio_thres", 1.2))
        self._device = device

    @staticmethod
    def _descriptors(keys):
        d = keys.descriptor_matrix if isinstance(keys, KeypointList) else keys
        d = np.ascontiguousarray(d, dtype=np.float32)
        if d.ndim != 2:
            raise ValueError("descriptors must be an N x dim matrix")
        return d

    def compute_matches(self):
        d1, d2 = self._d1, self._d2
        if d1.shape[0] and d2.shape[0] and d1.shape[1] != d2.shape[1]:
            raise ValueError("descriptor dimensions differ")
        dim = d1.shape[1] if d1.shape[0] else (d2.shape[1] if d2.ndim == 2 else 0)
        lib = capi.load()
        cap = 2 * (d1.shape[0] + d2.shape[0]) + 16
        for _ in range(2):
            out = np.zeros(max(cap, 1), MATCH_DTYPE)
            count = C.c_int(0)
            if self._self_matching:
                st = lib.sara_hip_self_match_descriptors(
                    d1.ctypes.data, self._regions.ctypes.data, d1.shape[0], dim,
                    self._ratio, self._metric, self._pixel, 0, out.ctypes.data,
                    cap, C.byref(count), self._device)
            else:
                st = lib.sara_hip_match_descriptors(
                    d1.ctypes.data, d1.shape[0], d2.ctypes.data, d2.shape[0], dim,
                    self._ratio, 0, out.ctypes.data, cap, C.byref(count),
                    self._device)
            if st == capi.CAPACITY_EXCEEDED and count.value > cap:
                cap = count.value       # several matches per key: exact count
                continue
            capi.check(st)
            break
        return out[:count.value]

    compute_self_matches = compute_matches


def match(keys1, keys2, lowe_ratio, device=0):
    """SfM/Helpers/KeypointMatching.cpp:19-25."""
    return AnnMatcher(keys1, keys2, lowe_ratio, device=device).compute_matches()


def _match_batch(pairs, dim, lowe_ratio, on_device, device, cap):
    lib = capi.load()
    arr = (capi.MatchPairStruct * max(len(pairs), 1))()
    for k, (p1, n1, p2, n2) in enumerate(pairs):
        arr[k] = capi.MatchPairStruct(p1, p2, n1, n2)
    offsets = (C.c_int * (len(pairs) + 1))()
    for _ in range(2):
        out = np.zeros(max(cap, 1), MATCH_DTYPE)
        st = lib.sara_hip_match_descriptors_batch(
            arr, len(pairs), dim, float(lowe_ratio), int(on_device),
            out.ctypes.data, cap, offsets, device)
        if st == capi.CAPACITY_EXCEEDED and offsets[len(pairs)] > cap:
            cap = offsets[len(pairs)]
            continue
        capi.check(st)
        break
    off = list(offsets)
    return [out[off[k]:off[k + 1]] for k in range(len(pairs))]


def match_pairs(key_pairs, lowe_ratio, device=0):
    """match() for a stream of independent pairs in ONE call
    (sara_hip_match_descriptors_batch): ``key_pairs`` = [(keys1, keys2), ...],
    KeypointLists or N x dim matrices; returns the list of match arrays, each
    byte-identical to ``match(keys1, keys2, lowe_ratio)``.  For ratios <= 1
    four searches are kept in flight on the device."""
    mats, pairs, dim = [], [], None
    for k1, k2 in key_pairs:
        d1, d2 = AnnMatcher._descriptors(k1), AnnMatcher._descriptors(k2)
        if d1.shape[1] != d2.shape[1] or (dim is not None and d1.shape[1] != dim):
            raise ValueError("descriptor dimensions differ")
        dim = d1.shape[1]
        mats.append((d1, d2))   # keep the arrays alive
        pairs.append((d1.ctypes.data, d1.shape[0], d2.ctypes.data, d2.shape[0]))
    if not pairs:
        return []
    cap = sum(2 * (n1 + n2) + 16 for _, n1, _, n2 in pairs)
    return _match_batch(pairs, dim, lowe_ratio, 0, device, cap)


def _ostream_float(v):
    """std::ostream << float with the default precision (6): %g."""
    return "%g" % float(np.float32(v))


def _eigen_block(values, rows, cols):
    """Eigen's operator<< with the default IOFormat (Eigen/src/Core/IO.h): one
    width for the whole expression = the widest coefficient, every coefficient
    right-aligned to it, " " between columns, newline between rows.  ``values``
    in column-major (storage) order.  Pinned on the 578 2 x 2 blocks of the
    reference's examples/Sara/Features/test.dogkey
    (tests/test_keypoint_text_pins.py)."""
    txt = [_ostream_float(v) for v in values]
    width = max((len(t) for t in txt), default=0)
    return "\n".join(" ".join(txt[c * rows + r].rjust(width)
                              for c in range(cols)) for r in range(rows))


def _eigen_row(values):
    """A row expression through Eigen's operator<<."""
    values = list(values)
    return _eigen_block(values, 1, len(values))


class H5File:
    """Core/HDF5.hpp:160-177 for keypoint files: a file name and how to open it
    ("w" = H5F_ACC_TRUNC, "a" = read-write, "r" = read-only).  Used with
    write_keypoints(h5_file, group_name, keys, overwrite) and
    read_keypoints(h5_file, group_name) - Features/IO.hpp:146-167.  Every call
    opens and closes the file (libsara_keypoint_h5.so over libhdf5)."""

    def __init__(self, filename, flags="r"):
        if flags not in ("r", "w", "a"):
            raise ValueError("flags must be 'r', 'w' or 'a'")
        self.filename, self.flags = str(filename), flags
        self._truncate_pending = flags == "w"
        _h5lib()

    def _write(self, group, keys, overwrite):
        if self.flags == "r":
            raise RuntimeError("file opened read-only")
        regions = np.ascontiguousarray(keys.regions, dtype=OEREGION_DTYPE)
        desc = np.ascontiguousarray(keys.descriptor_matrix, dtype=np.float32)
        if desc.ndim != 2 or desc.shape[0] != len(regions):
            raise ValueError("descriptors must be N x dim")
        st = _h5lib().sara_h5_write_keypoints(
            self.filename.encode(), int(self._truncate_pending), group.encode(),
            regions.ctypes.data, len(regions), desc.ctypes.data, desc.shape[1],
            int(bool(overwrite)))
        if st:
            raise RuntimeError(_h5lib().sara_h5_last_error().decode())
        self._truncate_pending = False

    def _read(self, group):
        lib = _h5lib()
        n, dim = C.c_int(), C.c_int()
        if lib.sara_h5_keypoints_sizes(self.filename.encode(), group.encode(),
                                       C.byref(n), C.byref(dim)):
            raise RuntimeError(lib.sara_h5_last_error().decode())
        regions = np.zeros(n.value, OEREGION_DTYPE)
        desc = np.zeros((n.value, dim.value), np.float32)
        if lib.sara_h5_read_keypoints(self.filename.encode(), group.encode(),
                                      regions.ctypes.data, desc.ctypes.data):
            raise RuntimeError(lib.sara_h5_last_error().decode())
        return KeypointList(regions, desc)


_h5 = None


def _h5lib():
    global _h5
    if _h5 is None:
        import os
        path = os.path.join(os.path.dirname(capi.LIB_PATH), "libsara_keypoint_h5.so")
        if not os.path.exists(path):
            raise RuntimeError(
                path + " is not built (needs libhdf5's C headers: "
                "make -C sara_amd/csrc)")
        lib = C.CDLL(path)
        lib.sara_h5_write_keypoints.argtypes = [
            C.c_char_p, C.c_int, C.c_char_p, C.c_void_p, C.c_int, C.c_void_p,
            C.c_int, C.c_int]
        lib.sara_h5_keypoints_sizes.argtypes = [
            C.c_char_p, C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        lib.sara_h5_read_keypoints.argtypes = [C.c_char_p, C.c_char_p, C.c_void_p,
                                               C.c_void_p]
        lib.sara_h5_last_error.restype = C.c_char_p
        _h5 = lib
    return _h5


def write_keypoints(features, descriptors, name, overwrite=False):
    """Features/IO.hpp:110-143: "N dim", then per keypoint
    ``x y m00 m10 m01 m11 orientation type d0 ... d(dim-1)`` (shape matrix in
    storage = column-major order).  ``features``: KeypointList.regions-like
    structured array; ``descriptors``: N x dim.

    HDF5 overload (Features/IO.hpp:159-167):
    ``write_keypoints(h5_file, group_name, keys, overwrite=False)``."""
    if isinstance(features, H5File):
        return features._write(descriptors, name, overwrite)
    if isinstance(features, KeypointList):
        descriptors = features.descriptor_matrix if descriptors is None \
            else descriptors
        features = features.regions
    d = np.asarray(descriptors, dtype=np.float32)
    try:
        f = open(name, "w", newline="\n")
    except OSError:
        return False
    with f:
        f.write("%d %d\n" % (len(features), d.shape[1] if d.ndim == 2 else 0))
        for i, r in enumerate(features):
            f.write("%s %s %s %s %d %s\n" % (
                _ostream_float(r["coords"][0]), _ostream_float(r["coords"][1]),
                _eigen_row(r["shape_matrix"]), _ostream_float(r["orientation"]),
                int(r["type"]), _eigen_row(d[i])))
    return True


def read_keypoints(name, group_name=None):
    """Features/IO.hpp:77-108 -> KeypointList.  As in the reference, the four
    shape coefficients are read ROW-major (OERegion's operator>>,
    Features/Feature.cpp:88-95 with Core/EigenExtension.hpp:163-170) although
    they were written in storage order.

    HDF5 overload (Features/IO.hpp:146-157):
    ``read_keypoints(h5_file, group_name)``."""
    if isinstance(name, H5File):
        return name._read(group_name)
    with open(name) as f:
        tok = f.read().split()
    n, dim = int(tok[0]), int(tok[1])
    regions = np.zeros(n, OEREGION_DTYPE)
    # fields the format does not carry keep OERegion's defaults
    regions["extremum_type"] = -2
    desc = np.zeros((n, dim), np.float32)
    at = 2
    for i in range(n):
        v = tok[at:at + 8 + dim]
        at += 8 + dim
        regions["coords"][i] = (np.float32(v[0]), np.float32(v[1]))
        m00, m01, m10, m11 = (np.float32(x) for x in v[2:6])
        regions["shape_matrix"][i] = (m00, m10, m01, m11)  # column-major storage
        regions["orientation"][i] = np.float32(v[6])
        regions["type"][i] = int(v[7])
        desc[i] = np.array(v[8:8 + dim], dtype=np.float32)
    return KeypointList(regions, desc)


def gradient_polar_coordinates(src, device=0):
    """Orientation.cpp:24-56 -> H x W x 2 (2*|grad|, atan2)."""
    s, sp = _f32(src)
    d = np.zeros(s.shape + (2,), np.float32)
    capi.check(capi.load().sara_hip_gradient_polar_coordinates(
        sp, s.shape[1], s.shape[0], d.ctypes.data_as(C.POINTER(C.c_float)),
        device))
    return d


def scale_space_dog_extremum_map(a, b, c, edge_ratio_thres=10.0,
                                 extremum_thres=0.01, img_padding_sz=1,
                                 device=0):
    """RefineExtremum.cpp:407-437 -> int8 map (+1 max, -1 min, 0)."""
    a, ap = _f32(a)
    b, bp = _f32(b)
    c, cp = _f32(c)
    out = np.zeros(a.shape, np.int8)
    capi.check(capi.load().sara_hip_scale_space_dog_extremum_map(
        ap, bp, cp, a.shape[1], a.shape[0], edge_ratio_thres, extremum_thres,
        img_padding_sz, out.ctypes.data_as(C.POINTER(C.c_int8)), device))
    return out
