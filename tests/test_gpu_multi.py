"""-m gpu: the native RCCL gather of the C-ABI (include/sara_hip_sift.h,
"Multi-GPU") on the one device a test box has: a single-process group of one
device (ncclCommInitAll) and a one-rank communicator (ncclCommInitRank) must
deliver exactly what detect() + fetch() returns, in frame order.  The N > 1
arithmetic (shard ranges, global offsets, empty ranks) is covered on the host
by tests/test_distributed_gloo.py and tests/test_capi_host.py."""
import numpy as np
import pytest

import sara_amd
from sara_amd import distributed as sd
from sara_amd.synth import synth_batch

pytestmark = pytest.mark.gpu


def params():
    return sara_amd.ImagePyramidParams(0, 6, num_octaves_max=3)


def reference(frames):
    with sara_amd.SiftContext(frames.shape[2], frames.shape[1], len(frames),
                              params()) as ctx:
        ctx.detect(frames)
        return ctx.fetch()


def test_single_process_group_of_one_device():
    frames = synth_batch(200, 160, 5)
    counts, regions, desc, so = reference(frames)
    g = sd.SiftGroup(200, 160, 5, params(), n_dev=1)
    try:
        for _ in range(2):                      # the second round reuses buffers
            res = g.detect(frames).gather(root=0)
            assert res.counts == [int(counts.sum())] and res.total == len(regions)
            f, d, s = res.host()
            assert f.tobytes() == regions.tobytes()
            assert np.array_equal(d, desc) and np.array_equal(s, so)
        u8 = np.round(frames * 255).astype(np.uint8)
        res8 = g.detect(u8).gather(root=0, with_descriptors=False)
        assert res8.total > 0 and res8.d_descriptors is None
        with pytest.raises(sara_amd.SaraHipError):
            g.gather(root=0)                    # nothing new to gather
        with pytest.raises(sara_amd.SaraHipError):
            g.detect(frames).gather(root=1)     # no such device in the group
    finally:
        g.close()


def test_one_rank_communicator():
    frames = synth_batch(200, 160, 3, first_index=7)
    counts, regions, desc, so = reference(frames)
    with sara_amd.SiftContext(200, 160, 3, params()) as ctx:
        comm = sd.Comm(ctx, sd.Comm.unique_id(), world_size=1, rank=0)
        try:
            t0 = ctx.submit(frames)
            t1 = ctx.submit(frames[::-1].copy())   # a second batch in flight
            res = comm.gather(t0, root=0)
            assert res.counts == [len(regions)]
            f, d, s = res.host()
            assert f.tobytes() == regions.tobytes()
            assert np.array_equal(d, desc) and np.array_equal(s, so)
            with pytest.raises(sara_amd.SaraHipError):
                ctx.collect(t0)                     # the gather consumed it
            off, r1, _, _ = ctx.collect(t1)
            assert int(off[-1]) == len(r1) > 0
        finally:
            comm.close()
