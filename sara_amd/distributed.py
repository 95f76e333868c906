"""Multi-GPU side of the SIFT front-end: frames are independent
(compute_sift_keypoints is a pure function of one image, SIFT.cpp:27-108), so
ranks shard them in contiguous blocks with no data-path collective.  The only
exchange is the gather of the variable-length keypoint arrays to the root:
counts first (all_gather), then one grouped send/recv per peer - a gatherv,
which RCCL does not provide natively.  On the GPU boxes the backend is "nccl"
(= RCCL over xGMI: every peer reaches the root over its own link); the same
code runs over "gloo" on CPU tensors in the tests.
"""
import ctypes as C
import os

import numpy as np

# torch only serves the torch.distributed variant below and is imported when
# that variant is first used: `import torch` makes the ROCm runtime it bundles
# the process's HIP runtime, and a process that only drives the library (the
# native communicator further down, the GPU tests) should keep the system's.
torch = dist = None


def _need_torch():
    global torch, dist
    if torch is None:
        import torch as _torch
        import torch.distributed as _dist
        torch, dist = _torch, _dist

from . import capi


def shard_range(n_frames, world_size, rank):
    """Contiguous block of frames of `rank`: frame f goes to rank
    floor(f * world_size / n_frames)."""
    lo = (rank * n_frames + world_size - 1) // world_size
    hi = ((rank + 1) * n_frames + world_size - 1) // world_size
    return lo, hi


class PendingGather:
    """Handle of an asynchronous gatherv: keeps the send/receive buffers alive
    until wait() returns."""

    def __init__(self, outs, counts, reqs, keep):
        self.outs, self.counts, self._reqs, self._keep = outs, counts, reqs, keep

    def wait(self):
        for r in self._reqs:
            r.wait()
        self._reqs, self._keep = [], None
        return self.outs, self.counts


def exchange_counts(n, group=None):
    """Every rank's keypoint count, exchanged through HOST tensors on `group`
    (a gloo group): unlike a device-side all_gather this neither waits for the
    kernels already enqueued on the GPU nor adds a device synchronisation."""
    _need_torch()
    world = dist.get_world_size(group)
    n_t = torch.tensor([int(n)], dtype=torch.int64)
    all_n = [torch.zeros_like(n_t) for _ in range(world)]
    dist.all_gather(all_n, n_t, group=group)
    return [int(t.item()) for t in all_n]


def gatherv_to_root(arrays, root=0, group=None, async_op=False, counts=None):
    """arrays: list of tensors whose first dimension is this rank's keypoint
    count n_r (same trailing shapes and dtypes on every rank).  Returns, on the
    root, (list of concatenated tensors in rank order, counts per rank); on the
    other ranks (None, counts per rank).  With async_op=True the point-to-point
    transfers are only posted and a PendingGather is returned, so the next
    batch's kernels overlap the exchange.  `counts` (from exchange_counts)
    skips the all_gather of the counts on the arrays' device."""
    _need_torch()
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)   # group-local; `root` is group-local as well
    n = int(arrays[0].shape[0])
    for a in arrays:
        assert int(a.shape[0]) == n
    dev = arrays[0].device

    def peer(r):
        """P2POp addresses peers by GLOBAL rank: translate a group-local one."""
        if group is None:
            return r
        return dist.get_global_rank(group, r)

    if counts is None:
        n_t = torch.tensor([n], device=dev, dtype=torch.int64)
        all_n = [torch.zeros_like(n_t) for _ in range(world)]
        dist.all_gather(all_n, n_t, group=group)
        counts = [int(t.item()) for t in all_n]
    else:
        counts = [int(c) for c in counts]
        assert len(counts) == world and counts[rank] == n
    if world == 1:
        if async_op:
            return PendingGather(list(arrays), counts, [], None)
        return list(arrays), counts

    if rank == root:
        total = sum(counts)
        outs = [torch.empty((total,) + tuple(a.shape[1:]), dtype=a.dtype,
                            device=dev) for a in arrays]
        offs, at = [], 0
        for r in range(world):
            offs.append(at)
            at += counts[r]
        for o, a in zip(outs, arrays):
            o[offs[root]:offs[root] + n] = a
        ops = []
        for r in range(world):
            if r == root or counts[r] == 0:
                continue
            for o in outs:
                ops.append(dist.P2POp(dist.irecv, o[offs[r]:offs[r] + counts[r]],
                                      peer(r), group))
        reqs = dist.batch_isend_irecv(ops) if ops else []
        if async_op:
            return PendingGather(outs, counts, reqs, (outs, arrays))
        for req in reqs:
            req.wait()
        return outs, counts
    ops = []
    sends = [a.contiguous() for a in arrays]
    if n:
        ops = [dist.P2POp(dist.isend, a, peer(root), group) for a in sends]
    reqs = dist.batch_isend_irecv(ops) if ops else []
    if async_op:
        return PendingGather(None, counts, reqs, sends)
    for req in reqs:
        req.wait()
    return None, counts


# --------------------------------------------------------------------------- #
# Native gather: RCCL inside the library (include/sara_hip_sift.h, "Multi-GPU").
# torch is not involved in the data path; a process-per-GPU launcher only has
# to ship the 128-byte communicator id between its processes.
# --------------------------------------------------------------------------- #
def shard_range_native(n_frames, world_size, rank):
    lo, hi = C.c_int(), C.c_int()
    capi.load().sara_hip_shard_range(n_frames, world_size, rank, C.byref(lo),
                                     C.byref(hi))
    return lo.value, hi.value


def _read_device(ptr, count, dtype, shape, device):
    out = np.zeros(shape, dtype)
    if count and ptr:
        capi.check(capi.load().sara_hip_copy_to_host(
            out.ctypes.data, ptr, out.nbytes, device))
    return out


class GatherResult:
    """Keypoints of the whole job on the root device (device pointers owned by
    the communicator) with the per-rank counts; host() copies them out."""

    def __init__(self, counts, d_feat, d_desc, d_so, total, device):
        self.counts, self.total, self.device = counts, total, device
        self.d_features, self.d_descriptors, self.d_scale_octave = d_feat, d_desc, d_so

    def host(self):
        from . import OEREGION_DTYPE
        n = self.total
        return (_read_device(self.d_features, n, OEREGION_DTYPE, (n,), self.device),
                _read_device(self.d_descriptors, n, np.float32, (n, 128), self.device),
                _read_device(self.d_scale_octave, n, np.int32, (n, 2), self.device))


class Comm:
    """One rank of a process-per-GPU gather group (sara_hip_comm_*)."""

    ID_BYTES = 128

    @staticmethod
    def unique_id():
        buf = (C.c_ubyte * Comm.ID_BYTES)()
        capi.check(capi.load().sara_hip_comm_unique_id(buf))
        return bytes(buf)

    def __init__(self, ctx, comm_id, world_size, rank, device=None):
        self.ctx, self.world_size, self.rank = ctx, world_size, rank
        self.device = ctx.device if device is None else device
        self._h = C.c_void_p()
        buf = (C.c_ubyte * Comm.ID_BYTES).from_buffer_copy(comm_id)
        capi.check(capi.load().sara_hip_comm_create(
            ctx._h, buf, world_size, rank, self.device, C.byref(self._h)))

    @property
    def transport(self):
        return capi.load().sara_hip_comm_transport(self._h).decode()

    def gather(self, ticket, root=0, with_descriptors=True):
        counts = (C.c_int * self.world_size)()
        f, d, s = C.c_void_p(), C.c_void_p(), C.c_void_p()
        total = C.c_int(0)
        capi.check(capi.load().sara_hip_comm_gather(
            self._h, ticket, root, 1 if with_descriptors else 0, counts,
            C.byref(f), C.byref(d), C.byref(s), C.byref(total)))
        getattr(self.ctx, "_inflight", {}).pop(ticket, None)
        return GatherResult(list(counts), f.value, d.value, s.value, total.value,
                            self.device)

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            try:
                capi.load().sara_hip_comm_destroy(self._h)
            except (AttributeError, TypeError):
                pass
            self._h = C.c_void_p()

    __del__ = close


class SiftGroup:
    """One process, one host thread per GPU (sara_hip_sift_group_*): the call
    pattern of a C++ consumer.  Frames are sharded in contiguous blocks."""

    def __init__(self, max_width, max_height, max_batch_per_device,
                 pyramid_params=None, gauss_truncate=4.0, extremum_thres=0.01,
                 edge_ratio_thres=10.0, extremum_refinement_iter=5,
                 max_keypoints=0, n_dev=1, devices=None):
        from . import ImagePyramidParams
        lib = capi.load()
        capi.require_gpu()
        self.params = pyramid_params or ImagePyramidParams()
        sp = capi.SiftParamsStruct(self.params._s, gauss_truncate, extremum_thres,
                                   edge_ratio_thres, int(extremum_refinement_iter))
        devs = (C.c_int * n_dev)(*devices) if devices is not None else None
        visible = max(lib.sara_hip_device_count(), 1)
        loop = os.environ.get("SARA_HIP_COMM_TRANSPORT") == "loopback"
        self.devices = (list(devices) if devices is not None else
                        [i % visible if loop else i for i in range(n_dev)])
        self.n_dev = n_dev
        self._h = C.c_void_p()
        capi.check(lib.sara_hip_sift_group_create(
            C.byref(sp), max_width, max_height, max_batch_per_device,
            max_keypoints, n_dev, devs, C.byref(self._h)))

    def detect(self, frames, last_stage=capi.STAGE_DESCRIPTOR):
        """frames: float32 [N, H, W] or uint8 [N, H, W] / [N, H, W, 3] host
        array; device i gets the i-th contiguous block."""
        a = np.ascontiguousarray(frames)
        if a.dtype == np.uint8:
            channels = 3 if a.ndim == 4 else 1
        else:
            a = np.ascontiguousarray(a, dtype=np.float32)
            channels = 0
        n, h, w = a.shape[:3]
        ptrs = (C.c_void_p * self.n_dev)()
        batch = (C.c_int * self.n_dev)()
        frame_bytes = a[0].nbytes
        self.shards = []
        for i in range(self.n_dev):
            lo, hi = shard_range_native(n, self.n_dev, i)
            self.shards.append((lo, hi))
            ptrs[i] = a.ctypes.data + lo * frame_bytes if hi > lo else None
            batch[i] = hi - lo          # 0: fewer frames than devices
        self._keepalive = a
        capi.check(capi.load().sara_hip_sift_group_detect(
            self._h, ptrs, batch, 0, channels, w, h, 0, int(last_stage)))
        return self

    def gather(self, root=0, with_descriptors=True):
        counts = (C.c_int * self.n_dev)()
        f, d, s = C.c_void_p(), C.c_void_p(), C.c_void_p()
        total = C.c_int(0)
        capi.check(capi.load().sara_hip_sift_group_gather(
            self._h, root, 1 if with_descriptors else 0, counts, C.byref(f),
            C.byref(d), C.byref(s), C.byref(total)))
        return GatherResult(list(counts), f.value, d.value, s.value, total.value,
                            self.devices[root])

    @property
    def transport(self):
        """"rccl", or "loopback" (SARA_HIP_COMM_TRANSPORT=loopback: the ranks are
        threads of this process - the N > 1 exchange on a one-GPU box)."""
        return capi.load().sara_hip_sift_group_transport(self._h).decode()

    def collect_host(self, with_descriptors=True, copy=True):
        """SURVEY.md section 8d's ending for a node: every device copies its
        shard into ONE pinned host array at its global offset over its own PCIe
        link.  -> (counts per device, regions, descriptors or None,
        scale_octave); views of group-owned pinned memory unless ``copy``."""
        from . import OEREGION_DTYPE
        counts = (C.c_int * self.n_dev)()
        f, d, s = C.c_void_p(), C.c_void_p(), C.c_void_p()
        total = C.c_int(0)
        capi.check(capi.load().sara_hip_sift_group_collect_host(
            self._h, 1 if with_descriptors else 0, counts, C.byref(f),
            C.byref(d), C.byref(s), C.byref(total)))
        n = total.value

        def view(ptr, dtype, shape):
            if n == 0 or not ptr.value:
                return np.zeros(shape, dtype)
            nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
            raw = np.frombuffer((C.c_char * nbytes).from_address(ptr.value),
                                dtype=np.uint8)
            if copy:
                raw = raw.copy()
            return raw.view(dtype).reshape(shape)

        return (list(counts), view(f, OEREGION_DTYPE, (n,)),
                view(d, np.float32, (n, 128)) if with_descriptors else None,
                view(s, np.int32, (n, 2)))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            try:
                capi.load().sara_hip_sift_group_destroy(self._h)
            except (AttributeError, TypeError):
                pass
            self._h = C.c_void_p()

    __del__ = close
