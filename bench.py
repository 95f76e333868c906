#!/usr/bin/env python
"""Benchmark of the MI355X SIFT front-end (BASELINE.json's metric).

  python bench.py --gpus N --steps K --warmup W

A *step* is one pass of the full SIFT hot path (Gaussian pyramid + DoG,
extrema + refinement, polar gradients, dominant orientations, 128-D
descriptors) over one batch of synthetic 1920x1080 frames per GPU, 4 octaves x
3 scales/octave, with the frames already resident in HBM when the timed region
starts.  Frames are independent, so ranks shard them with no data-path
collective; the only exchange is the gather of the variable-length keypoint
arrays to rank 0 over RCCL at the end of every step (N > 1).

Rank 0 prints ONE JSON line (see the driver's contract) that also carries
  "roofline"     : the Gaussian-pyramid kernels against the HBM roofline
  "cpu_baseline" : the CPU oracle timed on the host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames-per-gpu", type=int, default=64)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--octaves", type=int, default=4)
    ap.add_argument("--unique-frames", type=int, default=4,
                    help="distinct synthetic frames generated per rank (the "
                         "rest are their flips)")
    ap.add_argument("--cpu-frames", type=int, default=24,
                    help="frames of the cpu_baseline sample, about 10 s of CPU work "
                         "(0 = skip)")
    ap.add_argument("--stage", type=int, default=5,
                    help="last pipeline stage (5 = full SIFT)")
    return ap.parse_args()


def pyramid_bytes_per_frame(width, height, octaves, scales=6):
    """Algorithmic HBM bytes of the Gaussian-pyramid stage per frame
    (SURVEY.md 8d): every one of the `scales` planes of an octave is produced
    by one 4-byte read + one 4-byte write per pixel => 48*P for 6 planes per
    octave (the initial blur and the octave hand-overs are the s = 0 planes).
    The DoG pyramid is not materialised by this design, so it contributes no
    bytes here (the reference's separate DoG pass would add 60*P)."""
    total = 0
    w, h = width, height
    launches = 0
    for _ in range(octaves):
        total += 8 * w * h * scales
        launches += scales
        w //= 2
        h //= 2
    return total, launches


def pyramid_launches_per_step(width, height, octaves, batch, scales=6):
    """Kernel launches of the pyramid stage: one blur per plane (the first
    plane of octave 0 included); the hand-over to the next octave is fused into
    the blur that produces its source plane when that launch takes the marching
    path (>= 4 Mpx per launch, rows of 4 floats), else it is one more launch."""
    min_pixels = int(os.environ.get("SARA_HIP_MARCH_MIN_PIXELS", 4 << 20))
    n, w, h = 0, width, height
    for o in range(octaves):
        n += scales if o == 0 else scales - 1
        marching = w % 4 == 0 and w * h * batch >= min_pixels
        if o + 1 < octaves and not marching:
            n += 1
        w //= 2
        h //= 2
    return n


def host_cores():
    """CPUs this process may actually use: affinity mask capped by the cgroup
    CPU quota (the GPU boxes expose 256 logical CPUs under a 16-CPU quota;
    running OpenMP on all of them is 100x slower than on the quota)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_baseline(args, frames):
    """The CPU oracle (a port of Sara's CPU SIFT with the reference's loop
    structure and OpenMP pragmas, see oracle/sift_ref.hpp) on a bounded
    sample of the same workload, on this host's cores."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import refbind as rb
    rb.build()
    cores = host_cores()
    rb.lib().ref_omp_set_threads(cores)
    params = rb.PyramidParams(0, 6, None, 1, 0.5, 1.6, args.octaves)
    n = min(args.cpu_frames, len(frames))
    kp = 0
    stage = {}
    t0 = time.perf_counter()
    for i in range(n):
        r = rb.RefSift(frames[i], params, parallel=True)
        kp += len(r.keypoints()[0])
        for k, v in r.times().items():
            stage[k] = stage.get(k, 0.0) + v / n
    dt = time.perf_counter() - t0
    # the same port on one thread, one frame (SURVEY.md 8d asks for both)
    rb.lib().ref_omp_set_threads(1)
    t1 = time.perf_counter()
    kp1 = len(rb.RefSift(frames[0], params, parallel=True).keypoints()[0])
    dt1 = time.perf_counter() - t1
    rb.lib().ref_omp_set_threads(cores)
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {
        "value": kp / dt, "unit": "keypoints/s", "cores": cores, "kind": "port",
        "cpu_model": model,
        "value_single_thread": kp1 / dt1,
        "sample": "%d synthetic %dx%d frames, full SIFT, %d octaves; %.2f s "
                  "wall; OpenMP on the reference's pragmas" %
                  (n, args.width, args.height, args.octaves, dt),
        "ms_per_frame": 1e3 * dt / n,
        "stage_ms_per_frame": {k: round(v, 2) for k, v in stage.items()},
    }


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs `python -m torch.distributed.run "
                             "--nproc-per-node %d bench.py ...`" %
                             (args.gpus, args.gpus))
        args.gpus = world

    import torch
    import torch.distributed as dist
    import sara_amd
    from sara_amd import capi
    from sara_amd.synth import synth_batch
    from sara_amd.distributed import exchange_counts, gatherv_to_root

    ndev = capi.require_gpu()
    # SARA_BENCH_BACKEND=gloo lets the multi-process path be exercised on a box
    # with fewer GPUs than ranks (ranks then share devices and the gather is
    # staged through host memory); the real runs use nccl (= RCCL).
    backend = os.environ.get("SARA_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % ndev
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    # keypoint counts travel through host tensors on a gloo group: no wait for
    # the kernels in flight, no extra device synchronisation per step
    count_group = None
    if world > 1 and backend == "nccl":
        try:
            count_group = dist.new_group(backend="gloo")
        except Exception as e:  # no host-side group: counts go over RCCL
            count_group = None
            if rank == 0:
                print("gloo side group unavailable (%r): keypoint counts are "
                      "exchanged on the device" % (e,), file=sys.stderr)
        # establish the RCCL communicator collectively before the first
        # point-to-point exchange
        dist.all_reduce(torch.zeros(1, device=dev))
        torch.cuda.synchronize()

    B, W, H = args.frames_per_gpu, args.width, args.height
    frames_host = synth_batch(W, H, B, first_index=rank * B,
                              unique=args.unique_frames)
    frames = torch.from_numpy(frames_host).to(dev)  # resident in HBM
    torch.cuda.synchronize()

    params = sara_amd.ImagePyramidParams(0, 6, num_octaves_max=args.octaves)
    ctx = sara_amd.SiftContext(W, H, B, params, device=local_rank)
    stream = torch.cuda.current_stream(dev)

    # Keypoint arrays are gathered on rank 0.  Step i stages its results in
    # fresh device tensors (one device-to-device copy on the detect stream);
    # their exchange is posted right after the kernels of step i+1 have been
    # enqueued, so the GPU never waits for the host-side part of the gather
    # and the transfers overlap the next step's compute.
    feat_buf = desc_buf = so_buf = None
    pending = [None]   # point-to-point transfers in flight
    staged = [None]    # results staged on the device, exchange not posted yet

    def step():
        """One pass over the batch; returns this rank's keypoint count."""
        ctx.detect_device(frames.data_ptr(), B, W, H, last_stage=args.stage,
                          stream=stream.cuda_stream)
        if world > 1 and staged[0] is not None:
            post_gather(staged[0])
            staged[0] = None
        if args.stage < 4:
            c, _, _ = ctx.extrema() if args.stage >= 2 else (np.zeros(B), 0, 0)
            return int(np.sum(c))
        counts, total = ctx.counts()
        if world > 1:
            staged[0] = stage_results(total)
        return total

    def stage_results(total):
        """OERegion[ ] (48 B), descriptors (512 B) and (s,o) pairs of this step
        copied out of the context's buffers (which the next detect() reuses)."""
        mine_f = torch.empty((total, 48), dtype=torch.uint8, device=dev)
        mine_d = torch.empty((total, 128), dtype=torch.float32, device=dev)
        mine_s = torch.empty((total, 2), dtype=torch.int32, device=dev)
        if total:
            capi.check(capi.load().sara_hip_sift_fetch(
                ctx._h, mine_f.data_ptr(), mine_d.data_ptr(), mine_s.data_ptr(),
                1))  # asynchronous, on the detect stream
        return [mine_f, mine_d, mine_s], total

    def post_gather(item):
        """gatherv to rank 0 (sara_amd/distributed.py).  RCCL orders the
        transfers after everything already enqueued on the detect stream."""
        tensors, total = item
        if backend != "nccl":
            tensors = [t.cpu() for t in tensors]  # gloo: host staging
        # host-side exchange (gloo group, or the default group when that is
        # gloo); without one gatherv_to_root all_gathers them on the device
        host_counts = count_group is not None or backend != "nccl"
        counts_all = exchange_counts(total, count_group) if host_counts else None
        wait_gather()
        pending[0] = gatherv_to_root(tensors, root=0, async_op=True,
                                     counts=counts_all)

    def wait_gather():
        nonlocal feat_buf, desc_buf, so_buf
        if pending[0] is not None:
            outs, _ = pending[0].wait()
            pending[0] = None
            if outs is not None:
                feat_buf, desc_buf, so_buf = outs

    def sync():
        if world > 1:
            if staged[0] is not None:
                post_gather(staged[0])
                staged[0] = None
            wait_gather()
        torch.cuda.synchronize()
        ctx.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    stage_ms = {}
    kp_local = 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        kp_local += step()
        if ctx is not None and args.stage >= 1:
            for k, v in ctx.stage_times().items():
                stage_ms[k] = stage_ms.get(k, 0.0) + v
    sync()
    elapsed = time.perf_counter() - t0

    rdev = dev if backend == "nccl" else torch.device("cpu")
    el = torch.tensor([elapsed], device=rdev, dtype=torch.float64)
    kp = torch.tensor([kp_local], device=rdev, dtype=torch.int64)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(kp, op=dist.ReduceOp.SUM)
    elapsed = float(el.item())
    kp_total = int(kp.item())

    if rank == 0:
        steps = max(args.steps, 1)
        stage_ms = {k: v / steps for k, v in stage_ms.items()}
        bytes_frame, launches = pyramid_bytes_per_frame(W, H, args.octaves)
        pyr_launches = pyramid_launches_per_step(W, H, args.octaves, B)
        pyr_ms = stage_ms.get("pyramid", 0.0)
        achieved = (bytes_frame * B / 1e9) / (pyr_ms / 1e3) if pyr_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pyramid_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("hbm_bytes_per_step")
            except Exception:
                traffic = None
        out = {
            "metric": "SIFT keypoints/sec @1080p (4 octaves, 3 scales/oct); "
                      "1/2/4/8 GPU",
            "value": kp_total / elapsed,
            "unit": "keypoints/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "full SIFT (pyramid+DoG, extrema+refine, polar "
                            "gradients, orientations, 128-D descriptors) on "
                            "%d synthetic %dx%d frames per GPU, %d octaves x 3 "
                            "scales/octave, frames resident in HBM" %
                            (B, W, H, args.octaves),
                "frames_per_gpu": B,
                "global_frames": B * world,
                "keypoints_per_frame": kp_total / (steps * B * world),
                "frames_per_s": steps * B * world / elapsed,
                "parallelism": "frames sharded %d/GPU, RCCL gatherv of "
                               "keypoints to rank 0" % B if world > 1
                               else "single GPU",
                "last_stage": args.stage,
            },
            "stage_ms_per_step": {k: round(v, 4) for k, v in stage_ms.items()},
            "roofline": {
                "kernel": "gaussian_blur_march{,2}_kernel<R> (Gaussian pyramid "
                          "stage = %d launches/step on per-octave "
                          "streams; the octave hand-overs are fused into "
                          "the marching blurs; achieved = 48*P*frames / stage "
                          "time from HIP events)" % pyr_launches,
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "launches_per_step": pyr_launches,
                "avg_launch_us": 1e3 * pyr_ms / max(pyr_launches, 1),
                "algorithmic_bytes_per_step": bytes_frame * B,
                "us_per_frame": 1e3 * pyr_ms / B,
            },
        }
        if world == 1 and args.cpu_frames > 0:
            out["cpu_baseline"] = cpu_baseline(args, frames_host)
            out["config"]["gpu_over_cpu"] = out["value"] / max(
                out["cpu_baseline"]["value"], 1e-9)
        print(json.dumps(out))
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
