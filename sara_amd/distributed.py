"""Multi-GPU side of the SIFT front-end: frames are independent
(compute_sift_keypoints is a pure function of one image, SIFT.cpp:27-108), so
ranks shard them in contiguous blocks with no data-path collective.  The only
exchange is the gather of the variable-length keypoint arrays to the root:
counts first (all_gather), then one grouped send/recv per peer - a gatherv,
which RCCL does not provide natively.  On the GPU boxes the backend is "nccl"
(= RCCL over xGMI: every peer reaches the root over its own link); the same
code runs over "gloo" on CPU tensors in the tests.
"""
import torch
import torch.distributed as dist


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
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n = int(arrays[0].shape[0])
    for a in arrays:
        assert int(a.shape[0]) == n
    dev = arrays[0].device
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
                                      r, group))
        reqs = dist.batch_isend_irecv(ops) if ops else []
        if async_op:
            return PendingGather(outs, counts, reqs, (outs, arrays))
        for req in reqs:
            req.wait()
        return outs, counts
    ops = []
    sends = [a.contiguous() for a in arrays]
    if n:
        ops = [dist.P2POp(dist.isend, a, root, group) for a in sends]
    reqs = dist.batch_isend_irecv(ops) if ops else []
    if async_op:
        return PendingGather(None, counts, reqs, sends)
    for req in reqs:
        req.wait()
    return None, counts
