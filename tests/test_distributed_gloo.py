"""The N > 1 path on CPU: two gloo processes shard a list of frames, produce
real keypoint arrays (with the CPU oracle standing in for the GPU, which is
not available here) and gather them to rank 0 with the same gatherv code the
RCCL benchmark uses.  Root must end up with every frame's keypoints, in frame
order."""
import os
import socket

import numpy as np
import pytest

from sara_amd.distributed import (exchange_counts, gatherv_to_root,
                                  shard_range)

# torch is imported where it is used, not when pytest collects this module: the
# same pytest process runs the GPU tests (-m gpu), and `import torch` would make
# the ROCm runtime torch bundles that process's HIP runtime (sara_amd.distributed
# imports it lazily for the same reason).


def test_shard_range_partitions():
    for n in (1, 2, 7, 64, 512):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                lo, hi = shard_range(n, world, r)
                assert 0 <= lo <= hi <= n
                seen += list(range(lo, hi))
            assert seen == list(range(n))
    assert shard_range(512, 8, 3) == (192, 256)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_frames, tmpdir):
    import sys
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import refbind as rb
    from sara_amd.synth import synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(n_frames, world, rank)
    params = rb.PyramidParams(0, 6, None, 1, 0.5, 1.6, 2)
    regs, descs, sos, per_frame = [], [], [], []
    for f in range(lo, hi):
        img = synth(96, 80, 1234 + f)
        if f == 1:
            img[:] = 0.5  # a frame with no keypoints at all
        r, so, d = rb.RefSift(img, params).keypoints()
        regs.append(r.view(np.uint8).reshape(-1, 48))
        descs.append(d)
        sos.append(so)
        per_frame.append(len(r))
    cat = lambda xs, shape, dt: (np.concatenate(xs) if xs else
                                 np.zeros(shape, dt))
    arrays = [torch.from_numpy(cat(regs, (0, 48), np.uint8)),
              torch.from_numpy(cat(descs, (0, 128), np.float32)),
              torch.from_numpy(cat(sos, (0, 2), np.int32))]
    # blocking and asynchronous forms must agree
    outs, counts = gatherv_to_root(arrays, root=0)
    pending = gatherv_to_root(arrays, root=0, async_op=True)
    outs2, counts2 = pending.wait()
    assert counts2 == counts
    # the bench's form: counts exchanged on a separate (gloo) group first,
    # several exchanges in flight one after the other
    side = dist.new_group(backend="gloo")
    counts3 = exchange_counts(arrays[0].shape[0], side)
    assert counts3 == counts
    p3 = gatherv_to_root(arrays, root=0, async_op=True, counts=counts3)
    p4 = gatherv_to_root(arrays, root=0, async_op=True,
                         counts=exchange_counts(arrays[0].shape[0], side))
    outs3, _ = p3.wait()
    outs4, _ = p4.wait()
    if rank == 0:
        assert all(torch.equal(a, b) for a, b in zip(outs, outs2))
        assert all(torch.equal(a, b) for a, b in zip(outs, outs3))
        assert all(torch.equal(a, b) for a, b in zip(outs, outs4))
    assert counts[rank] == sum(per_frame)
    if rank == 0:
        np.savez(os.path.join(tmpdir, "root.npz"), regs=outs[0].numpy(),
                 desc=outs[1].numpy(), so=outs[2].numpy(),
                 counts=np.array(counts))
    else:
        assert outs is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_frames", [(2, 5), (2, 1)])
def test_gatherv_over_gloo(oracle, tmp_path, world, n_frames):
    port = _free_port()
    import torch.multiprocessing as mp
    mp.start_processes(_worker, args=(world, port, n_frames, str(tmp_path)),
                       nprocs=world, join=True, start_method="spawn")
    got = np.load(tmp_path / "root.npz")
    from sara_amd.synth import synth
    params = oracle.PyramidParams(0, 6, None, 1, 0.5, 1.6, 2)
    regs, descs, sos = [], [], []
    for f in range(n_frames):
        img = synth(96, 80, 1234 + f)
        if f == 1:
            img[:] = 0.5
        r, so, d = oracle.RefSift(img, params).keypoints()
        regs.append(r.view(np.uint8).reshape(-1, 48))
        descs.append(d)
        sos.append(so)
    assert np.array_equal(got["regs"], np.concatenate(regs))
    assert np.array_equal(got["desc"], np.concatenate(descs))
    assert np.array_equal(got["so"], np.concatenate(sos))
    assert int(got["counts"].sum()) == len(got["regs"]) > 0


def _subgroup_worker(rank, world, port):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sub = dist.new_group(ranks=[1, 2])       # group-local 0, 1 = global 1, 2
    if rank in (1, 2):
        local = dist.get_rank(sub)
        n = 3 + local
        arrays = [torch.full((n, 4), float(rank)),
                  torch.arange(n, dtype=torch.int32).reshape(n, 1) + 100 * rank]
        outs, counts = gatherv_to_root(arrays, root=0, group=sub)
        assert counts == [3, 4]
        if local == 0:                       # the root of the SUBGROUP is global 1
            assert outs[0].shape == (7, 4)
            assert torch.equal(outs[0][:3], torch.full((3, 4), 1.0))
            assert torch.equal(outs[0][3:], torch.full((4, 4), 2.0))
            assert outs[1].flatten().tolist() == [100, 101, 102, 200, 201, 202, 203]
        else:
            assert outs is None
    dist.barrier()
    dist.destroy_process_group()


def test_gatherv_in_a_subgroup():
    """Peers of P2POp are GLOBAL ranks: a gather inside a subgroup whose ranks
    are not 0..n-1 must translate its group-local ranks."""
    import torch.multiprocessing as mp
    mp.start_processes(_subgroup_worker, args=(3, _free_port()), nprocs=3,
                       join=True, start_method="spawn")
