"""-m gpu: the native RCCL gather of the C-ABI (include/sara_hip_sift.h,
"Multi-GPU") on the one device a test box has: a single-process group of one
device (ncclCommInitAll) and a one-rank communicator (ncclCommInitRank) must
deliver exactly what detect() + fetch() returns, in frame order; the N > 1
exchange runs on the same device through the loopback transport (below)."""
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


# --------------------------------------------------------------------------- #
# N > 1 on ONE device: the loopback transport (SARA_HIP_COMM_TRANSPORT=loopback)
# runs the ranks as threads of this process and moves the bytes with device
# copies, so every branch of the gatherv in sara_amd/csrc/sift_comm.cpp - global
# offsets, empty ranks, root != 0, the header AllGather, failing ranks - runs
# here.  What it cannot show is RCCL / xGMI itself: that is the driver's
# multi-GPU run.  Contract: SURVEY.md section 8e; frames are independent
# (FeatureDetectors/SIFT.cpp:27-108), so the root must hold exactly what one
# context returns for the concatenated frames.
# --------------------------------------------------------------------------- #
import threading


@pytest.fixture
def loopback(monkeypatch):
    monkeypatch.setenv("SARA_HIP_COMM_TRANSPORT", "loopback")
    monkeypatch.setenv("SARA_HIP_LOOPBACK_TIMEOUT_MS", "8000")


@pytest.mark.parametrize("n_dev,n_frames,root", [
    (2, 5, 0), (2, 5, 1), (3, 7, 1), (3, 2, 2),      # 3 ranks, 2 frames: one empty
    (8, 16, 5), (8, 5, 7), (8, 5, 0)])                # 8 ranks, 5 frames: 3 empty
def test_loopback_group_gather_equals_single_context(loopback, n_dev, n_frames, root):
    frames = synth_batch(200, 160, n_frames, first_index=3 * n_dev + root)
    counts, regions, desc, so = reference(frames)
    per_dev = (n_frames + n_dev - 1) // n_dev
    g = sd.SiftGroup(200, 160, per_dev, params(), n_dev=n_dev)
    try:
        assert g.transport == "loopback"
        res = g.detect(frames).gather(root=root)
        # per-rank counts = sums over the rank's contiguous shard
        off = np.concatenate([[0], np.cumsum(counts)])
        want = [int(off[hi] - off[lo]) for lo, hi in g.shards]
        assert res.counts == want and res.total == len(regions)
        if n_frames < n_dev:
            assert 0 in [hi - lo for lo, hi in g.shards]     # an empty rank took part
        f, d, s = res.host()
        assert f.tobytes() == regions.tobytes()
        assert d.tobytes() == desc.tobytes() and s.tobytes() == so.tobytes()
        # a second round with another root reuses / regrows the buffers
        root2 = (root + 1) % n_dev
        res = g.detect(frames[::-1].copy()).gather(root=root2, with_descriptors=False)
        c2, r2, _, s2 = reference(frames[::-1].copy())
        f, _, s = res.host()
        assert res.d_descriptors is None
        assert f.tobytes() == r2.tobytes() and s.tobytes() == s2.tobytes()
        # host delivery: every device into one pinned array at its offset
        hc, hf, hd, hs = g.detect(frames).collect_host()
        assert hc == want
        assert hf.tobytes() == regions.tobytes()
        assert hd.tobytes() == desc.tobytes() and hs.tobytes() == so.tobytes()
    finally:
        g.close()


def test_loopback_group_failure_paths_release_the_batch(loopback):
    frames = synth_batch(200, 160, 4)
    _, regions, desc, _ = reference(frames)
    g = sd.SiftGroup(200, 160, 2, params(), n_dev=2)
    try:
        g.detect(frames)
        with pytest.raises(sara_amd.SaraHipError):
            g.gather(root=2)                    # rejected before any state change
        res = g.gather(root=1)                  # ... so the batch is still there
        assert res.host()[0].tobytes() == regions.tobytes()
        with pytest.raises(sara_amd.SaraHipError):
            g.gather(root=0)                    # consumed
        # descriptors of a batch that stopped at the orientation stage: every
        # rank refuses, nobody blocks, and the tickets are released - so more
        # than two further detect() calls still work
        g.detect(frames, last_stage=sara_amd.STAGE_ORIENTATION)
        with pytest.raises(sara_amd.SaraHipError) as e:
            g.gather(root=0, with_descriptors=True)
        assert e.value.status == sara_amd.capi.NOT_READY
        for _ in range(3):
            g.detect(frames)                    # never gathered: dropped by the next
        res = g.gather(root=0)
        assert res.host()[1].tobytes() == desc.tobytes()
    finally:
        g.close()


def _run_ranks(n, fn):
    out, err = [None] * n, [None] * n

    def work(r):
        try:
            out[r] = fn(r)
        except Exception as e:   # noqa: BLE001 - reported by the caller
            err[r] = e

    ts = [threading.Thread(target=work, args=(r,)) for r in range(n)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(60)
    assert not any(t.is_alive() for t in ts), "a rank is blocked in the gather"
    return out, err


@pytest.mark.parametrize("world,root", [(2, 0), (3, 2), (4, 1)])
def test_loopback_process_per_gpu_form(loopback, world, root):
    """sara_hip_comm_* (the torchrun layout, one communicator per rank) with the
    ranks as threads: header AllGather, root growth handshake, transfers."""
    n_frames = 2 * world + 1
    frames = synth_batch(200, 160, n_frames, first_index=40 + world)
    counts, regions, desc, so = reference(frames)
    comm_id = sd.Comm.unique_id()
    assert comm_id.startswith(b"SARA-LOOPBACK-ID")
    ctxs = [sara_amd.SiftContext(200, 160, 3, params()) for _ in range(world)]
    comms = [None] * world
    try:
        def first(r):
            comms[r] = sd.Comm(ctxs[r], comm_id, world, r)
            assert comms[r].transport == "loopback"
            lo, hi = sd.shard_range_native(n_frames, world, r)
            t = ctxs[r].submit(frames[lo:hi])
            res = comms[r].gather(t, root=root)
            return res, (res.host() if r == root else None)

        out, err = _run_ranks(world, first)
        assert err == [None] * world, err
        off = np.concatenate([[0], np.cumsum(counts)])
        want = [int(off[sd.shard_range(n_frames, world, r)[1]] -
                    off[sd.shard_range(n_frames, world, r)[0]]) for r in range(world)]
        for r in range(world):
            assert out[r][0].counts == want           # filled on every rank
            assert (out[r][0].d_features is not None) == (r == root)
        f, d, s = out[root][1]
        assert f.tobytes() == regions.tobytes()
        assert d.tobytes() == desc.tobytes() and s.tobytes() == so.tobytes()

        # one rank fails locally (its ticket was already consumed): every rank
        # must come back with an error instead of waiting for it
        def second(r):
            lo, hi = sd.shard_range_native(n_frames, world, r)
            t = ctxs[r].submit(frames[lo:hi])
            if r == world - 1:
                ctxs[r].collect(t)
            return comms[r].gather(t, root=root)

        out, err = _run_ranks(world, second)
        assert all(isinstance(e, sara_amd.SaraHipError) for e in err), err
        assert err[world - 1].status == sara_amd.capi.NOT_READY
        # ... and the tickets of the healthy ranks were released: two more
        # batches go through
        for _ in range(2):
            out, err = _run_ranks(world, first)
            assert err == [None] * world, err
        assert out[root][1][0].tobytes() == regions.tobytes()
    finally:
        for c in comms:
            if c is not None:
                c.close()
        for c in ctxs:
            c.close()


def test_collect_into_caller_memory_and_stage_check():
    """sara_hip_sift_ticket_counts / _collect_into: the read-back into memory the
    caller owns (what lets several processes fill one shared host array), and
    collect()'s refusal to hand out descriptors that were never computed."""
    frames = synth_batch(200, 160, 3, first_index=11)
    counts, regions, desc, so = reference(frames)
    with sara_amd.SiftContext(200, 160, 3, params()) as ctx:
        t = ctx.submit(frames)
        off, total = ctx.ticket_counts(t)
        assert total == len(regions) and list(np.diff(off)) == list(counts)
        pad = 7                                   # land at an offset, like a shard
        f = np.zeros(total + pad, sara_amd.OEREGION_DTYPE)
        d = np.zeros((total + pad, 128), np.float32)
        s = np.zeros((total + pad, 2), np.int32)
        lib = sara_amd.capi.load()
        sara_amd.capi.check(lib.sara_hip_host_register(d.ctypes.data, d.nbytes))
        try:
            ctx.collect_into(t, f[pad:].ctypes.data, d[pad:].ctypes.data,
                             s[pad:].ctypes.data)
        finally:
            sara_amd.capi.check(lib.sara_hip_host_unregister(d.ctypes.data))
        assert f[pad:].tobytes() == regions.tobytes()
        assert d[pad:].tobytes() == desc.tobytes() and s[pad:].tobytes() == so.tobytes()
        assert not d[:pad].any()
        with pytest.raises(sara_amd.SaraHipError):
            ctx.collect(t)                        # consumed
        t = ctx.submit(frames, last_stage=sara_amd.STAGE_ORIENTATION)
        with pytest.raises(sara_amd.SaraHipError) as e:
            ctx.collect(t, with_descriptors=True)
        assert e.value.status == sara_amd.capi.NOT_READY
        _, r2, d2, _ = ctx.collect(t, with_descriptors=False)   # still pending
        assert d2 is None and len(r2) == total


# --------------------------------------------------------------------------- #
# bench.py --gpus N as the driver starts it: `python bench.py --gpus N ...`
# with no launcher and no WORLD_SIZE.  On the one-GPU test box N = 2 exceeds the
# device count, so the run goes through the single-process group form over the
# loopback transport and must SAY so; with a launcher (--launch procs) the two
# ranks share the GPU, RCCL refuses, and the line must say gloo - never "rccl".
# --------------------------------------------------------------------------- #
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*argv, timeout=900):
    env = dict(os.environ)
    env.pop("SARA_HIP_COMM_TRANSPORT", None)
    env.pop("SARA_HIP_MARCH_MIN_PIXELS", None)     # the shipped kernel selection
    env.pop("SARA_HIP_STRIP_GROUP", None)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv),
                       capture_output=True, text=True, timeout=timeout, env=env,
                       cwd=ROOT)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert p.returncode == 0 and len(lines) == 1, (p.returncode, p.stdout[-2000:],
                                                   p.stderr[-4000:])
    return json.loads(lines[0]), p.stderr


def test_bench_two_gpus_without_a_launcher():
    line, err = _bench("--gpus", "2", "--steps", "2", "--warmup", "1", "--no-extras")
    assert line["n_gpus"] == 2 and line["steps"] == 2
    assert line["metric"].startswith("SIFT keypoints/sec @1080p")
    assert line["config"]["global_frames"] == 128
    assert line["gather_verified"] is True
    assert line["value"] > 0 and line["ms_per_step"] > 0
    import sara_amd.capi as capi
    if capi.load().sara_hip_device_count() >= 2:
        assert line["transport"] == "rccl" and line["rccl_nranks"] == 2
        assert line["rccl_version"]
    else:
        # one device: honest about not being an RCCL run
        assert line["transport"] == "loopback" and line["rccl_nranks"] == 0
        assert "LOOPBACK" in line["config"]["parallelism"]
        assert "LOOPBACK" in err
    # 2 x 64 frames of ~4.4 k keypoints each
    assert 3500 < line["config"]["keypoints_per_frame"] < 5500


def test_bench_two_ranks_under_the_launcher():
    line, err = _bench("--gpus", "2", "--steps", "2", "--warmup", "1", "--no-extras",
                       "--launch", "procs", "--frames-per-gpu", "4", "--width", "640",
                       "--height", "360")
    assert line["n_gpus"] == 2 and line["config"]["global_frames"] == 8
    import sara_amd.capi as capi
    if capi.load().sara_hip_device_count() >= 2:
        assert line["transport"] == "rccl" and line["rccl_nranks"] == 2
        assert line["gather_verified"] is True
    else:
        assert line["transport"] == "torch.distributed/gloo"
        assert line["rccl_nranks"] == 0 and "GLOO" in err


def test_bench_refuses_loopback_on_a_box_with_enough_devices():
    """SARA_HIP_COMM_TRANSPORT=loopback must not turn an N <= device_count run
    into a line that reads like RCCL: --gpus 1 is unaffected, and the group form
    refuses outright when the box has a device per rank."""
    env = dict(os.environ, SARA_HIP_COMM_TRANSPORT="loopback")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    import sara_amd.capi as capi
    ndev = capi.load().sara_hip_device_count()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus",
                        str(max(ndev, 2)), "--steps", "1", "--warmup", "0",
                        "--no-extras", "--launch", "group", "--frames-per-gpu", "2",
                        "--width", "320", "--height", "240"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    if ndev >= 2:
        assert p.returncode != 0 and "refusing" in p.stderr
    else:
        line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
        assert line["transport"] == "loopback"
