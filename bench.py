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
    ap.add_argument("--unique-frames", type=int, default=0,
                    help="distinct synthetic frames generated per rank (0 = "
                         "every frame has its own seed, 1234 + global index; "
                         "k > 0: k frames and their flips)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the secondary measurements (host-to-host rate, "
                         "single-image config 2, 4K config 5)")
    ap.add_argument("--strict-h2h", action="store_true",
                    help="N > 1: fail the run when the host-to-host (SURVEY 8d) "
                         "measurement cannot be made, instead of printing the "
                         "line with value_host_to_host_gray8 = null and the reason")
    ap.add_argument("--cpu-frames", type=int, default=24,
                    help="frames of the cpu_baseline sample, about 10 s of CPU work "
                         "(0 = skip)")
    ap.add_argument("--stage", type=int, default=5,
                    help="last pipeline stage (5 = full SIFT)")
    ap.add_argument("--launch", choices=("auto", "procs", "group"), default="auto",
                    help="N > 1 without a launcher (WORLD_SIZE unset): 'procs' "
                         "re-executes this script under torch.distributed.run, one "
                         "process per GPU; 'group' drives all GPUs from this one "
                         "process (sara_hip_sift_group_*, the C++ caller's form); "
                         "'auto' = procs when the box has >= N devices, else group "
                         "over the loopback test transport (said so in the line)")
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
    # one warm-up, then the median of >= 5 runs.  The warm-up is
    # one pass over all n sample frames - it also produces the oracle results
    # the benchmarked batch is checked against; each timed pass then covers the
    # first m frames (the sample is bounded to 10-30 s of CPU work in all).
    # Only the oracle's compute is timed (the RefSift constructor = the
    # reference's compute_sift_keypoints call), not the ctypes copy-out.
    results = []
    for i in range(n):
        results.append(rb.RefSift(frames[i], params, parallel=True).keypoints())
    m = min(8, n)
    passes = 5
    rates, pass_s = [], []
    stage = {}
    for _ in range(passes):
        kp, dt = 0, 0.0
        for i in range(m):
            t0 = time.perf_counter()
            r = rb.RefSift(frames[i], params, parallel=True)
            dt += time.perf_counter() - t0
            kp += rb.lib().ref_sift_keypoint_count(r._h)
            for k, v in r.times().items():
                stage[k] = stage.get(k, 0.0) + v / (m * passes)
            del r
        rates.append(kp / dt)
        pass_s.append(dt)
    order = sorted(range(passes), key=lambda j: rates[j])
    med = order[passes // 2]
    # the same port on one thread, one frame (SURVEY.md 8d asks for both):
    # the median of three runs after the warm-up above
    rb.lib().ref_omp_set_threads(1)
    single = []
    for _ in range(3):
        t1 = time.perf_counter()
        r = rb.RefSift(frames[0], params, parallel=True)
        dt1 = time.perf_counter() - t1
        single.append(rb.lib().ref_sift_keypoint_count(r._h) / dt1)
        del r
    rb.lib().ref_omp_set_threads(cores)
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return results, {
        "value": rates[med], "unit": "keypoints/s", "cores": cores, "kind": "port",
        "cpu_model": model,
        "value_single_thread": sorted(single)[1],
        "protocol": "1 warm-up pass over %d frames, then %d timed passes over "
                    "the first %d; value = median pass; only the oracle's "
                    "compute is timed" % (n, passes, m),
        "passes_keypoints_per_s": [round(v, 1) for v in rates],
        "sample": "%d synthetic %dx%d frames per pass, full SIFT, %d octaves; "
                  "median pass %.2f s wall; OpenMP on the reference's pragmas" %
                  (m, args.width, args.height, args.octaves, pass_s[med]),
        "ms_per_frame": 1e3 * pass_s[med] / m,
        "stage_ms_per_frame": {k: round(v, 2) for k, v in stage.items()},
    }


def check_parity(gpu, oracle_results):
    """The benchmarked batch against the oracle, frame by frame (outside the
    timed region): keypoint count, order, (s, o) pairs exact; coordinates and
    extremum values exact; shape rel 1e-6; orientation 1e-6 rad; descriptors
    max-abs 2e-3 on 0..255 - the bars of tests/test_gpu_pipeline.py.  Raises on
    the first mismatch; returns the number of frames checked."""
    counts, regions, desc, so = gpu
    off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    worst = 0.0
    for i, (rreg, rso, rdesc) in enumerate(oracle_results):
        sl = slice(int(off[i]), int(off[i + 1]))
        if int(counts[i]) != len(rreg):
            raise SystemExit("parity: frame %d has %d keypoints, oracle %d" %
                             (i, int(counts[i]), len(rreg)))
        g = regions[sl]
        ok = (np.array_equal(so[sl], rso) and
              np.array_equal(g["coords"], rreg["coords"]) and
              np.array_equal(g["extremum_value"], rreg["extremum_value"]) and
              np.array_equal(g["extremum_type"], rreg["extremum_type"]) and
              np.allclose(g["shape_matrix"], rreg["shape_matrix"], rtol=1e-6,
                          atol=0) and
              np.allclose(g["orientation"], rreg["orientation"], rtol=0,
                          atol=1e-6))
        err = float(np.max(np.abs(desc[sl] - rdesc))) if len(rreg) else 0.0
        worst = max(worst, err)
        if not ok or err > 2e-3:
            raise SystemExit("parity: frame %d differs from the oracle "
                             "(descriptor max-abs %.3g)" % (i, err))
    return len(oracle_results), worst


def timed(fn, steps, warmup):
    for _ in range(warmup):
        fn()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    return (time.perf_counter() - t0) / max(steps, 1)


def host_to_host(ctx, frames_host, args, torch):
    """SURVEY.md 8d's metric: frames in pinned host memory -> OERegion[] +
    descriptors in (pinned) host memory, two batches in flight (submit /
    collect).  Returns keypoints/s for float32 and for gray8 frames."""
    import sara_amd
    B, H, W = frames_host.shape
    out = {}
    for name, arr, ch in (
            ("float32", frames_host, 0),
            ("gray8", np.round(frames_host * 255.0).astype(np.uint8), 1)):
        pinned = torch.from_numpy(np.ascontiguousarray(arr)).pin_memory()
        ptr = pinned.data_ptr()
        # Two call orders: "stage first" enqueues the next
        # upload before the host waits for a read-back - the copy engine goes
        # from one upload straight into the next - but how the runtime then
        # places upload and read-back on the copy engines differs from process
        # to process (8.7-9.1 or 12.4 ms per float32 step); "submit first" is
        # the round-2 order.  Both are measured, the better one is reported.
        best = None
        for order in ("stage(i+1); collect(i-1); submit_staged(i+1)",
                      "submit(i+1); collect(i)"):
            state = {"tickets": [], "kp": 0}

            def step():
                if order.startswith("stage"):
                    ctx.stage_raw(ptr, ch, B, W, H)
                    if len(state["tickets"]) == 2:
                        off, _, _, _ = ctx.collect(state["tickets"].pop(0))
                        state["kp"] += int(off[-1])
                    state["tickets"].append(ctx.submit_staged())
                else:
                    state["tickets"].append(ctx.submit_raw(ptr, ch, B, W, H))
                    if len(state["tickets"]) == 2:
                        off, _, _, _ = ctx.collect(state["tickets"].pop(0))
                        state["kp"] += int(off[-1])

            for _ in range(4):  # grows the pinned result buffers of both slots
                step()
            steps = max(args.steps, 8)
            state["kp"] = 0
            t0 = time.perf_counter()
            for _ in range(steps):  # steady state: `steps` uploads and read-backs
                step()
            dt = time.perf_counter() - t0
            for t in state["tickets"]:
                ctx.collect(t)
            res = {"keypoints_per_s": state["kp"] / dt,
                   "ms_per_step": 1e3 * dt / steps, "call_order": order}
            if best is None:
                best = dict(res, ms_per_step_by_order={})
            if res["ms_per_step"] < best["ms_per_step"]:
                best.update(res)
            best["ms_per_step_by_order"][order] = res["ms_per_step"]
        out[name] = best
        del pinned
    return out


def host_to_host_system_runtime(args):
    """The same step in a subprocess WITHOUT torch (tools/h2h_notorch.py says
    why: torch's wheel loads its own, older HIP runtime, under which upload and
    read-back do not overlap on the copy engines).  None when it cannot run."""
    import subprocess
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools",
                          "h2h_notorch.py")
    try:
        res = subprocess.run(
            [sys.executable, script, "--json", "--steps", str(max(args.steps, 8)),
             "--frames", str(args.frames_per_gpu), "--width", str(args.width),
             "--height", str(args.height), "--octaves", str(args.octaves)],
            capture_output=True, text=True, timeout=300)
        line = [l for l in res.stdout.splitlines() if l.startswith("{")][-1]
        return json.loads(line)
    except Exception as e:  # noqa: BLE001 - a secondary measurement
        sys.stderr.write("host_to_host_system_runtime: %r\n" % (e,))
        return None


def pyramid_per_kernel(args, torch, dev, frames):
    """Per-launch device times of the Gaussian-pyramid stage on ONE stream
    (SARA_HIP_OPT_SINGLE_STREAM + SARA_HIP_OPT_LAUNCH_TIMERS: a hipEvent pair
    around every launch, on the stream it is launched on), same frames and
    batch as the timed region.  -> (per-kernel list, serial stage ms).  Each
    blur reads and writes every pixel of its plane once: 8 B per pixel."""
    import sara_amd
    from sara_amd import capi
    B, W, H = args.frames_per_gpu, args.width, args.height
    params = sara_amd.ImagePyramidParams(0, 6, num_octaves_max=args.octaves)
    acc = {}
    with sara_amd.SiftContext(W, H, B, params, device=dev.index or 0) as c:
        c.set_option(capi.OPT_SINGLE_STREAM, 1)
        c.set_option(capi.OPT_LAUNCH_TIMERS, 1)
        reps = 0
        for i in range(6):
            c.detect_device(frames.data_ptr(), B, W, H, last_stage=1)
            c.synchronize()
            if i < 2:
                continue
            reps += 1
            for r in c.pyramid_launches():
                k = (int(r["octave"]), int(r["scale"]), int(r["taps"]),
                     int(r["pixels"]))
                acc[k] = acc.get(k, 0.0) + float(r["ms"])
    rows, total_ms = [], 0.0
    for (o, sc, taps, px), ms in sorted(acc.items()):
        ms /= reps
        total_ms += ms
        rows.append({"octave": o, "scale": sc, "radius": taps // 2,
                     "us": round(1e3 * ms, 2),
                     "GBs": round(8 * px / 1e9 / (ms / 1e3), 1),
                     "frac": round(8 * px / 1e9 / (ms / 1e3) / HBM_PEAK_GBS, 4)})
    return rows, total_ms


def odd_width_case(torch, dev):
    """A width that is not a multiple of 4 (1366 x 768 video): such rows cannot
    be cut into float4 strips, so the pyramid takes another path.  Recorded so
    that the cost of that path is a number."""
    import sara_amd
    from sara_amd.synth import synth_batch
    out = {}
    p4 = sara_amd.ImagePyramidParams(0, 6, num_octaves_max=4)
    for name, W, H in (("1366x768", 1366, 768), ("1368x768", 1368, 768)):
        B = 64
        f = torch.from_numpy(synth_batch(W, H, B, unique=8)).to(dev)
        P = sum((W >> o) * (H >> o) for o in range(4))
        with sara_amd.SiftContext(W, H, B, p4, device=dev.index or 0) as c:
            pyr, tot, kp = [], [], 0
            for i in range(6):
                c.detect_device(f.data_ptr(), B, W, H)
                _, kp = c.counts()
                if i >= 2:
                    st = c.stage_times()
                    pyr.append(st["pyramid"])
                    tot.append(st["total"])
        pyr_ms, tot_ms = float(np.mean(pyr)), float(np.mean(tot))
        out[name] = {"pyramid_ms": pyr_ms, "ms_per_step": tot_ms,
                     "pyramid_frac_of_hbm_peak":
                         48 * P * B / 1e9 / (pyr_ms / 1e3) / HBM_PEAK_GBS,
                     "keypoints_per_s": kp / (tot_ms / 1e3)}
        del f
    out["workload"] = ("64 frames, 4 octaves, full SIFT; 1366 is not a multiple "
                       "of 4, 1368 is the nearest width that is")
    out["pyramid_slowdown"] = (out["1366x768"]["pyramid_ms"] /
                               out["1368x768"]["pyramid_ms"])
    return out


def match_config(torch, dev):
    """The immediate consumer: AnnMatcher / match() on the
    keypoints of two 1080p frames, descriptors resident in HBM."""
    import sara_amd
    from sara_amd.synth import synth_batch
    from sara_amd.synth import synth
    W, H = 1920, 1080
    p4 = sara_amd.ImagePyramidParams(0, 6, num_octaves_max=4)
    # two views of one scene: 1080p crops of a larger synthetic frame, shifted
    # by (24, 8) pixels - what consecutive video frames give the matcher
    scene = synth(W + 24, H + 8, 1234)
    frames = np.ascontiguousarray(
        np.stack([scene[:H, :W], scene[8:H + 8, 24:W + 24]]))
    out = {}
    with sara_amd.SiftContext(W, H, 2, p4, device=dev.index or 0) as c:
        c.detect(frames)
        counts, _ = c.counts()
        n1, n2 = int(counts[0]), int(counts[1])
        # straight through the C-ABI with the device pointers of the two frames'
        # descriptors (match_frames adds a counts() round trip per call)
        import ctypes as C
        from sara_amd import capi
        off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        _, d_desc, _, _ = c.device_results()
        cap = 64 * (n1 + n2)
        buf = np.zeros(cap, capi.MATCH_DTYPE)
        cnt = C.c_int(0)
        lib = capi.load()

        def call(ratio):
            capi.check(lib.sara_hip_match_descriptors(
                d_desc + int(off[0]) * 512, n1, d_desc + int(off[1]) * 512, n2, 128,
                ratio, 1, buf.ctypes.data, cap, C.byref(cnt), dev.index or 0))

        for name, ratio in (("ratio_0.6", 0.6), ("ratio_1.2_default", 1.2)):
            call(ratio)
            t = timed(lambda: call(ratio), 50, 5)
            out[name] = {"ms_per_pair": 1e3 * t, "matches": int(cnt.value)}
        # A stream of pairs in ONE call (sara_hip_match_descriptors_batch): the
        # same pair 16 times - the pair is a grid dimension of every kernel, one
        # tile grid over all pairs, one dense read-back.
        P = 16
        pairs = (capi.MatchPairStruct * P)()
        for k in range(P):
            pairs[k] = capi.MatchPairStruct(d_desc + int(off[0]) * 512,
                                            d_desc + int(off[1]) * 512, n1, n2)
        offs = (C.c_int * (P + 1))()
        bbuf = np.zeros(P * (n1 + n2), capi.MATCH_DTYPE)

        def batch():
            capi.check(lib.sara_hip_match_descriptors_batch(
                pairs, P, 128, 0.6, 1, bbuf.ctypes.data, len(bbuf), offs,
                dev.index or 0))

        batch()
        tb = timed(batch, 10, 2) / P
        same = all(bbuf[offs[k]:offs[k + 1]].tobytes() == bbuf[:offs[1]].tobytes()
                   for k in range(P))
        call(0.6)
        same = same and bbuf[:offs[1]].tobytes() == buf[:cnt.value].tobytes()
        out["batch_16_pairs_ratio_0.6"] = {
            "ms_per_pair": 1e3 * tb, "matches_per_pair": int(offs[1]),
            "identical_to_single_calls": bool(same),
            "frac_of_bf16_mfma_peak": 3 * 2.0 * n1 * n2 * 128 / tb / 2.5e15}
        # The consumer's unit of work (SfM/Helpers/KeypointMatching.cpp:19-25 after
        # OdometryPipeline::detect_keypoints): detect two frames, match them.  The
        # frames are resident in HBM, the keypoints never leave the device between
        # the two steps; the match list ends on the host.
        d_frames = torch.from_numpy(frames).to(dev)

        def pair(ratio):
            c.detect_device(d_frames.data_ptr(), 2, W, H)
            cts, _ = c.counts()                    # also orders the matcher behind detect
            o1 = int(cts[0])
            _, dd, _, _ = c.device_results()
            capi.check(lib.sara_hip_match_descriptors(
                dd, o1, dd + o1 * 512, int(cts[1]), 128, ratio, 1, buf.ctypes.data, cap,
                C.byref(cnt), dev.index or 0))

        fe = {}
        for name, ratio in (("ratio_0.6", 0.6), ("ratio_1.2_default", 1.2)):
            pair(ratio)
            fe[name] = {"ms_per_pair": 1e3 * timed(lambda: pair(ratio), 50, 5),
                        "matches": int(cnt.value)}
        t_det = timed(lambda: (c.detect_device(d_frames.data_ptr(), 2, W, H), c.counts()),
                      50, 5)
        fe["ms_detect_two_frames"] = 1e3 * t_det
        fe["workload"] = ("detect (one call, batch of two 1080p frames resident in HBM) "
                          "+ counts + AnnMatcher on the device-resident descriptors, "
                          "match list on the host: the front-end's work per image pair")
        out["front_end_pair"] = fe
        del d_frames
    out["workload"] = ("AnnMatcher::compute_matches on %d x %d SIFT descriptors "
                       "(two 1080p frames), both directions, descriptors in HBM, "
                       "match list on the host" % (n1, n2))
    # exact arithmetic of the exhaustive search: n1*n2*128 (sub, mul, add) per
    # direction; the prefilter path does one n1*n2*128 f32 MFMA contraction
    out["pair_distance_terms"] = 2 * n1 * n2 * 128
    # one pass over the n1 x n2 tiles = three bf16 products per pair of keys
    # (hi hi + hi lo + lo hi of the split rows) = 3 * 2 n1 n2 128 flop on the
    # matrix cores (ratios <= 1: one pass; the radius search of ratios > 1: two),
    # against the dense bf16 MFMA peak of /opt/skills/guides/MI355X_MICROARCH.md
    flop = 3 * 2.0 * n1 * n2 * 128
    out["mfma_flop_per_pass"] = flop
    out["mfma_passes"] = {"ratio_0.6": 1, "ratio_1.2_default": 2}
    out["default_ratio_tail"] = ("ranks, scores, (x, y) duplicates of the two "
                                 "directions and the final order on the device; one "
                                 "read-back of the finished list")
    out["mfma_time_at_peak_us_per_pass"] = flop / 2.5e15 * 1e6
    out["producer"] = ("MFMA prefilter (v_mfma_f32_32x32x16_bf16 on a hi / lo bf16 "
                       "split of the rows, both directions from one contraction, "
                       "rigorous error guard; tile minima carry their position, so "
                       "ratios <= 1 need one pass; a pass is staging- and "
                       "epilogue-bound, not MFMA-bound) + exact FLANN-order "
                       "re-ranking; identical lists to the exhaustive search")
    return out


def secondary_configs(args, torch, dev):
    """BASELINE.json configs 2 and 5 as secondary measurements (rank 0, N = 1):
    config 2 = ONE 1920x1080 frame, pyramid + DoG + extrema only, against the
    128*P algorithmic bytes of SURVEY.md 8d; the same frame through the full
    pipeline (the drop-in's per-call operating point); config 5 = 16 frames of
    3840x2160, 5 octaves, Gaussian-pyramid stage against 48*P."""
    import sara_amd
    from sara_amd.synth import synth_batch
    out = {}
    # ---- config 2 / single image
    W, H = 1920, 1080
    one = synth_batch(W, H, 1)
    d_one = torch.from_numpy(one).to(dev)
    p4 = sara_amd.ImagePyramidParams(0, 6, num_octaves_max=4)
    P = sum((W >> o) * (H >> o) for o in range(4))
    with sara_amd.SiftContext(W, H, 1, p4, device=dev.index or 0) as c1:
        def run(stage):
            c1.detect_device(d_one.data_ptr(), 1, W, H, last_stage=stage)
            c1.synchronize()
        t2 = timed(lambda: run(2), 200, 20)
        t5 = timed(lambda: run(5), 200, 20)
        c1.set_option(sara_amd.capi.OPT_STAGE_TIMERS, 1)
        h2h = timed(lambda: c1.collect(c1.submit(one)), 100, 10)
        u8 = np.round(one * 255.0).astype(np.uint8)
        h2h8 = timed(lambda: c1.collect(c1.submit(u8)), 100, 10)
        n1 = int(c1.collect(c1.submit(one))[0][-1])
        # a video stream: frame i + 1 is submitted before frame i is collected
        # (pinned frames; per-frame period, not the latency of one call)
        pin = {}
        for name, arr, ch in (("float32", one, 0), ("gray8", u8, 1)):
            pa = sara_amd.pinned_empty(arr.shape, arr.dtype)
            pa[...] = arr
            state = {"t": None}

            def step(pa=pa, ch=ch, state=state):
                t = c1.submit_raw(pa.ctypes.data, ch, 1, W, H)
                if state["t"] is not None:
                    c1.collect(state["t"])
                state["t"] = t

            pin[name] = timed(step, 200, 20)
            c1.collect(state["t"])
            # the latency of one call when the frame is decoded into pinned memory
            pin[name + "_latency"] = timed(
                lambda pa=pa, ch=ch: c1.collect(c1.submit_raw(pa.ctypes.data, ch, 1, W, H)),
                100, 10)
            del pa
    out["config2"] = {
        "workload": "1 x 1920x1080, pyramid + DoG + extrema only (stage 2), "
                    "frame resident in HBM, HIP-graph replay",
        "ms": 1e3 * t2,
        "algorithmic_bytes": 128 * P,
        "achieved_GBs": 128 * P / 1e9 / t2,
        "frac_of_hbm_peak": 128 * P / 1e9 / t2 / HBM_PEAK_GBS,
    }
    out["single_image"] = {
        "workload": "1 x 1920x1080 full SIFT per call (the reference's "
                    "compute_sift_keypoints call pattern)",
        "keypoints": n1,
        "ms_hbm_resident": 1e3 * t5,
        "ms_host_float32_to_host": 1e3 * h2h,
        "ms_host_gray8_to_host": 1e3 * h2h8,
        "ms_host_float32_pinned_to_host": 1e3 * pin["float32_latency"],
        "ms_host_gray8_pinned_to_host": 1e3 * pin["gray8_latency"],
        "ms_per_frame_two_in_flight_float32": 1e3 * pin["float32"],
        "ms_per_frame_two_in_flight_gray8": 1e3 * pin["gray8"],
        "two_in_flight": "submit(frame i+1) before collect(frame i), pinned host "
                         "frames: the period of a video stream, not the latency "
                         "of one call",
        "keypoints_per_s_host_to_host": n1 / h2h,
    }
    # ---- config 5: 4K, 5 octaves
    W4, H4, B4 = 3840, 2160, 16
    f4 = synth_batch(W4, H4, B4, unique=4)
    d4 = torch.from_numpy(f4).to(dev)
    p5 = sara_amd.ImagePyramidParams(0, 6, num_octaves_max=5)
    P4 = sum((W4 >> o) * (H4 >> o) for o in range(5))
    capi = sara_amd.capi
    sweep = []
    with sara_amd.SiftContext(W4, H4, B4, p5, device=dev.index or 0) as c4:
        def stage_ms(n_runs=8, stage=5):
            pyr, tot, kp = [], [], 0
            for i in range(n_runs):
                c4.detect_device(d4.data_ptr(), B4, W4, H4, last_stage=stage)
                _, kp = c4.counts() if stage >= 5 else (None, 0)
                if stage < 5:
                    c4.synchronize()
                if i >= 2:
                    st = c4.stage_times()
                    pyr.append(st["pyramid"])
                    tot.append(st["total"])
            return float(np.mean(pyr)), float(np.mean(tot)), kp
        c4.set_option(capi.OPT_KERNEL_SELECTION, capi.SELECT_SHIPPED)
        c4.set_option(capi.OPT_GRAPH_REPLAY, 0)  # per-stage timers: plain launches
        pyr_ms, tot_ms, kp = stage_ms()
        # BASELINE.json words config 5 as an LDS tile-size sweep.  Measured IN
        # THIS RUN (round 6: the kernel selection is a context option): the
        # shipped decomposition - marching strips, which have no 2-D LDS tile:
        # a wave owns 256 / 128 columns x a row segment - against the LDS-tiled
        # blur kernel in each tile shape it is compiled for, and the marching
        # kernels' own axis (waves per launch = segment height).  Pyramid stage
        # only (last_stage 1), 6 timed runs each.
        combos = [
            ("shipped: marching strips 256 / 128 columns, 4096 / 2048 waves "
             "per launch", {capi.OPT_KERNEL_SELECTION: capi.SELECT_SHIPPED}),
            ("LDS tiles 64 x 32 (512 threads)",
             {capi.OPT_KERNEL_SELECTION: capi.SELECT_TILED_BLUR,
              capi.OPT_TILE_GEOMETRY: 1}),
            ("LDS tiles 64 x 16 (256 threads)",
             {capi.OPT_KERNEL_SELECTION: capi.SELECT_TILED_BLUR,
              capi.OPT_TILE_GEOMETRY: 2}),
            ("LDS tiles 32 x 16 (128 threads)",
             {capi.OPT_KERNEL_SELECTION: capi.SELECT_TILED_BLUR,
              capi.OPT_TILE_GEOMETRY: 3}),
            ("marching, 1024 waves per launch",
             {capi.OPT_KERNEL_SELECTION: capi.SELECT_SHIPPED,
              capi.OPT_MARCH_WAVES: 1024}),
            ("marching, 8192 waves per launch",
             {capi.OPT_KERNEL_SELECTION: capi.SELECT_SHIPPED,
              capi.OPT_MARCH_WAVES: 8192}),
        ]
        for label, combo in combos:
            c4.set_option(capi.OPT_TILE_GEOMETRY, 0)
            c4.set_option(capi.OPT_MARCH_WAVES, 0)
            for opt in sorted(combo):  # the selection (10) first: it resets 11, 12
                c4.set_option(opt, combo[opt])
            ms, _, _ = stage_ms(8, stage=1)
            sweep.append({"decomposition": label, "pyramid_ms": round(ms, 3),
                          "achieved_GBs": round(48 * P4 * B4 / 1e9 / (ms / 1e3)),
                          "frac": round(48 * P4 * B4 / 1e9 / (ms / 1e3) /
                                        HBM_PEAK_GBS, 3)})
    out["config5"] = {
        "workload": "16 x 3840x2160, 5 octaves x 3 scales/octave, full SIFT; "
                    "pyramid stage from HIP events",
        "pyramid_ms": pyr_ms,
        "pyramid_achieved_GBs": 48 * P4 * B4 / 1e9 / (pyr_ms / 1e3),
        "pyramid_frac_of_hbm_peak":
            48 * P4 * B4 / 1e9 / (pyr_ms / 1e3) / HBM_PEAK_GBS,
        "ms_per_step": tot_ms,
        "keypoints_per_s": kp / (tot_ms / 1e3),
        "keypoints_per_frame": kp / B4,
        "sweep": sweep,
        "sweep_source": "measured in this run (SARA_HIP_OPT_KERNEL_SELECTION / "
                        "_TILE_GEOMETRY / _MARCH_WAVES on one context, pyramid "
                        "stage only); rocprofv3 occupancy per blur kernel of "
                        "the same decompositions: profiles/r05_4k_sweep.txt "
                        "(tools/sweep_4k.sh)",
    }
    # ---- opt-in fused-multiply-add blurs (SARA_HIP_OPT_FMA_BLUR): not bit-exact,
    # never part of `value` / `roofline`; reported for comparison only
    Wb, Hb, Bb = 1920, 1080, 64
    fb = synth_batch(Wb, Hb, Bb, unique=8)
    db = torch.from_numpy(fb).to(dev)
    Pb = sum((Wb >> o) * (Hb >> o) for o in range(4))
    with sara_amd.SiftContext(Wb, Hb, Bb, p4, device=dev.index or 0) as cb:
        res = {}
        for name, on in (("exact", 0), ("fma", 1)):
            cb.set_option(sara_amd.capi.OPT_FMA_BLUR, on)
            pyr, tot = [], []
            for i in range(8):
                cb.detect_device(db.data_ptr(), Bb, Wb, Hb)
                cb.counts()
                if i >= 2:
                    st = cb.stage_times()
                    pyr.append(st["pyramid"])
                    tot.append(st["total"])
            res[name] = (float(np.mean(pyr)), float(np.mean(tot)))
    out["fma_blur_option"] = {
        "note": "SARA_HIP_OPT_FMA_BLUR = 1 (opt-in, pyramids within 3e-7 of the "
                "range instead of bit-exact); same 64 x 1080p workload",
        "pyramid_ms_exact": res["exact"][0], "pyramid_ms_fma": res["fma"][0],
        "pyramid_frac_fma": 48 * Pb * Bb / 1e9 / (res["fma"][0] / 1e3) / HBM_PEAK_GBS,
        "ms_per_step_exact": res["exact"][1], "ms_per_step_fma": res["fma"][1],
    }
    del d_one, d4, db
    return out


def verify_gather(ctx, comm, frames, args, dist, rank, world):
    """N > 1, outside the timed region: the BYTES that reach the root, not just
    their count.  Every rank hashes its own results (detect + fetch of the very
    frames it benchmarks; the pipeline is deterministic), the hashes travel
    over gloo, and rank 0 compares them with the slices of one more RCCL
    gather at the per-rank offsets."""
    import hashlib
    B, W, H = args.frames_per_gpu, args.width, args.height
    ctx.detect_device(frames.data_ptr(), B, W, H, last_stage=5)
    _, reg, desc, so = ctx.fetch()
    mine = [hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
            for a in (reg, desc, so)] + [len(reg)]
    everyone = [None] * world
    dist.all_gather_object(everyone, mine)
    t = ctx.submit_raw(frames.data_ptr(), 0, B, W, H, on_device=True, last_stage=5)
    res = comm.gather(t, root=0)
    if rank != 0:
        return None
    f, d, s2 = res.host()
    at = 0
    for r, (hf, hd, hs, n) in enumerate(everyone):
        if res.counts[r] != n:
            raise SystemExit("gather: rank %d sent %d keypoints, the root has %d"
                             % (r, n, res.counts[r]))
        got = [hashlib.sha256(np.ascontiguousarray(a[at:at + n]).tobytes())
               .hexdigest() for a in (f, d, s2)]
        if got != [hf, hd, hs]:
            raise SystemExit("gather: the bytes of rank %d on the root differ "
                             "from what it computed" % r)
        at += n
    return {"ranks": world, "keypoints": at,
            "what": "sha256 of OERegion[], descriptors and (s,o) of every rank's "
                    "shard == sha256 of the root's slice at that rank's offset"}


def host_to_host_multi(ctx, frames_host, args, torch, dist, rank, world, kp_hint):
    """SURVEY.md 8d's ending for N > 1: results in HOST memory, without
    funnelling them through the root GPU (8 x 64 frames x 4.4 k keypoints x 568 B
    = 1.27 GB per step would cross ONE PCIe link: about 23 ms against a 7 ms
    step).  Every rank copies its shard straight into ONE host array - a shared
    mapping every process registers with HIP - at its global offset, over its
    own PCIe link; the counts that fix the offsets travel over gloo.  gray8
    frames in pinned host memory in, two batches in flight per rank."""
    from multiprocessing import shared_memory
    from sara_amd import capi
    B, H, W = frames_host.shape
    cap = int(1.3 * kp_hint * world) + 4096           # keypoints of one step
    sizes = (48 * cap, 512 * cap, 8 * cap)
    name = [None]
    shm = None
    if rank == 0:
        shm = shared_memory.SharedMemory(create=True, size=sum(sizes))
        name[0] = shm.name
    dist.broadcast_object_list(name, src=0)
    if rank != 0:
        shm = shared_memory.SharedMemory(name=name[0])
        try:  # only the creator unlinks; keep this rank's tracker out of it
            from multiprocessing import resource_tracker
            resource_tracker.unregister(shm._name, "shared_memory")
        except Exception:
            pass
    buf = np.frombuffer(shm.buf, dtype=np.uint8)
    base = buf.ctypes.data
    lib = capi.load()
    capi.check(lib.sara_hip_host_register(base, sum(sizes)))
    u8 = torch.from_numpy(np.round(frames_host * 255.0).astype(np.uint8)).pin_memory()
    state = {"ticket": None, "kp": 0}

    def deliver(ticket):
        _, total = ctx.ticket_counts(ticket)
        n_t = torch.tensor([total], dtype=torch.int64)
        all_n = [torch.zeros_like(n_t) for _ in range(world)]
        dist.all_gather(all_n, n_t)
        counts = [int(t.item()) for t in all_n]
        if sum(counts) > cap:
            raise SystemExit("host_to_host_multi: %d keypoints exceed the shared "
                             "array (%d)" % (sum(counts), cap))
        at = sum(counts[:rank])
        ctx.collect_into(ticket, base + 48 * at, base + sizes[0] + 512 * at,
                         base + sizes[0] + sizes[1] + 8 * at)
        return sum(counts)

    def step():
        t = ctx.submit_raw(u8.data_ptr(), 1, B, W, H, on_device=False)
        if state["ticket"] is not None:
            state["kp"] += deliver(state["ticket"])
        state["ticket"] = t

    for _ in range(3):
        step()
    deliver(state["ticket"])
    state.update(ticket=None, kp=0)
    steps = max(args.steps, 8)
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    state["kp"] += deliver(state["ticket"])
    dist.barrier()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    capi.check(lib.sara_hip_host_unregister(base))
    del buf
    shm.close()
    if rank == 0:
        shm.unlink()
    return {"keypoints_per_s": state["kp"] / float(dt.item()),
            "ms_per_step": 1e3 * float(dt.item()) / steps,
            "definition": "gray8 frames in pinned host memory on every rank -> "
                          "ONE host array (shared, HIP-registered) holding all "
                          "ranks' OERegion[] + descriptors + (s,o) in global "
                          "frame order; every GPU writes its shard over its own "
                          "PCIe link; counts over gloo"}


def rccl_version_string():
    import ctypes as C
    from sara_amd import capi
    v = C.c_int(0)
    if capi.load().sara_hip_rccl_version(C.byref(v)) != 0 or v.value <= 0:
        return None
    return "%d.%d.%d" % (v.value // 10000, v.value // 100 % 100, v.value % 100)


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_procs(args):
    """`python bench.py --gpus N` without a launcher: the same script again under
    torch.distributed.run, one process per GPU on this node (what the driver's
    own N > 1 command does).  Its rank 0 prints the line; the exit code is
    passed on."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get(
        "HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    sys.stderr.write("bench: --gpus %d without WORLD_SIZE: %s\n" %
                     (args.gpus, " ".join(cmd)))
    raise SystemExit(subprocess.call(cmd, env=env))


def run_group(args, ndev):
    """N GPUs driven by THIS process: sara_hip_sift_group_* (one host thread and
    one context per device, ncclCommInitAll, frames sharded in contiguous
    blocks, gatherv to device 0) - the form a C++ caller of Sara uses, no
    launcher and no torch.distributed.  On a box with fewer devices than N the
    ranks share the devices and the exchange runs over the library's loopback
    test transport; the line then says so (transport / parallelism) - that is a
    plumbing check, not a scaling measurement."""
    import ctypes as C
    import hashlib
    import torch
    import sara_amd
    from sara_amd import capi
    from sara_amd.distributed import SiftGroup
    from sara_amd.synth import synth_batch

    N, B, W, H = args.gpus, args.frames_per_gpu, args.width, args.height
    loopback = N > ndev
    if loopback:
        os.environ["SARA_HIP_COMM_TRANSPORT"] = "loopback"
        sys.stderr.write(
            "bench: --gpus %d on a box with %d device(s): the %d ranks share the "
            "device(s) and the gather runs over the LOOPBACK test transport, not "
            "RCCL\n" % (N, ndev, N))
    elif os.environ.get("SARA_HIP_COMM_TRANSPORT") == "loopback":
        raise SystemExit("SARA_HIP_COMM_TRANSPORT=loopback is set although the box "
                         "has %d >= %d devices: refusing to report a loopback run "
                         "as an RCCL one" % (ndev, N))
    devices = [i % ndev for i in range(N)]
    params = sara_amd.ImagePyramidParams(0, 6, num_octaves_max=args.octaves)
    # create() fails loudly (non-zero exit) when RCCL cannot form the group
    group = SiftGroup(W, H, B, params, n_dev=N, devices=devices)
    transport = group.transport
    if transport != ("loopback" if loopback else "rccl"):
        raise SystemExit("group transport is %r" % transport)
    lib = capi.load()
    frames_host = [synth_batch(W, H, B, first_index=r * B,
                               unique=args.unique_frames or None) for r in range(N)]
    frames = [torch.from_numpy(f).to(torch.device("cuda", devices[r]))
              for r, f in enumerate(frames_host)]
    for d in set(devices):
        torch.cuda.synchronize(d)
    ptrs = (C.c_void_p * N)(*[f.data_ptr() for f in frames])
    batch = (C.c_int * N)(*([B] * N))

    def step():
        capi.check(lib.sara_hip_sift_group_detect(
            group._h, ptrs, batch, 0, 0, W, H, 1, int(args.stage)))
        res = group.gather(root=0, with_descriptors=args.stage >= 5)
        return res

    for _ in range(args.warmup):
        step()
    for d in set(devices):
        torch.cuda.synchronize(d)
    kp_total = 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        kp_total += step().total       # gather() returns with the root's arrays complete
    for d in set(devices):
        torch.cuda.synchronize(d)
    elapsed = time.perf_counter() - t0

    # outside the timed region: the bytes on the root against every rank's own
    # detect() + fetch() (the pipeline is deterministic)
    res = step()
    f, dsc, so = res.host()
    at, verified = 0, args.stage >= 5
    for r in range(N if args.stage >= 5 else 0):
        with sara_amd.SiftContext(W, H, B, params, device=devices[r]) as c:
            c.detect_device(frames[r].data_ptr(), B, W, H, last_stage=5)
            _, reg, desc, rso = c.fetch()
        n = len(reg)
        same = (res.counts[r] == n and
                f[at:at + n].tobytes() == reg.tobytes() and
                dsc[at:at + n].tobytes() == desc.tobytes() and
                so[at:at + n].tobytes() == rso.tobytes())
        if not same:
            raise SystemExit("gather: the bytes of rank %d on the root differ from "
                             "what it computed" % r)
        at += n
    steps = max(args.steps, 1)
    out = {
        "metric": "SIFT keypoints/sec @1080p (4 octaves, 3 scales/oct); 1/2/4/8 GPU",
        "value": kp_total / elapsed, "unit": "keypoints/s", "n_gpus": N,
        "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / steps, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "transport": transport, "rccl_nranks": N if transport == "rccl" else 0,
        "rccl_version": rccl_version_string() if transport == "rccl" else None,
        "gather_verified": bool(verified),
        "launch": "one process, one host thread per GPU (sara_hip_sift_group_*)",
        "value_definition": (
            "HBM-resident: frames in HBM when the timed region starts, keypoint "
            "arrays of all ranks in the HBM of device 0 when it ends; detect and "
            "gather of a step are not overlapped with the next step in this form"),
        "config": {
            "workload": "full SIFT on %d synthetic %dx%d frames per GPU, %d octaves "
                        "x 3 scales/octave, frames resident in HBM" %
                        (B, W, H, args.octaves),
            "frames_per_gpu": B, "global_frames": B * N,
            "keypoints_per_frame": kp_total / (steps * B * N),
            "frames_per_s": steps * B * N / elapsed,
            "devices": devices, "devices_on_the_box": ndev,
            "parallelism": (
                "frames sharded %d/GPU over %d ranks, gatherv of keypoints to "
                "rank 0: %s" % (B, N,
                "LOOPBACK test transport (the ranks share %d device(s): plumbing "
                "check, not a scaling measurement)" % ndev if loopback else
                "library RCCL (ncclCommInitAll; grouped ncclSend/ncclRecv at the "
                "global offsets)")),
            "last_stage": args.stage,
        },
    }
    if verified:
        out["config"]["gather_verified"] = {
            "ranks": N, "keypoints": at,
            "what": "OERegion[], descriptors and (s,o) of every rank's shard on "
                    "the root == that rank's own detect() + fetch(), byte for byte"}
    print(json.dumps(out))
    group.close()


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher: start the N-GPU job from here.  torch first: its wheel
        # brings its own HIP runtime, which must be the one the process loads
        # before the library pulls in the image's (the other order leaves torch
        # without devices)
        import torch
        ndev = torch.cuda.device_count()
        if ndev < 1:
            raise SystemExit("bench: no HIP device")
        mode = args.launch
        if mode == "auto":
            mode = "procs" if ndev >= args.gpus else "group"
        if mode == "procs":
            launch_procs(args)
        return run_group(args, ndev)
    if world != args.gpus:
        args.gpus = world

    import torch
    import torch.distributed as dist
    import sara_amd
    from sara_amd import capi
    from sara_amd.synth import synth_batch
    from sara_amd.distributed import exchange_counts, gatherv_to_root

    ndev = capi.require_gpu()
    # Gather of the keypoint arrays on rank 0 (N > 1):
    #   SARA_BENCH_GATHER=rccl (default): the library's own RCCL gatherv
    #     (sara_hip_comm_*: counts by ncclAllGather, grouped ncclSend/ncclRecv);
    #     torch.distributed (gloo) only ships the 128-byte communicator id and
    #     the final timing reduction - it is not on the data path.
    #   SARA_BENCH_GATHER=torch: the torch.distributed variant of
    #     sara_amd/distributed.py (backend SARA_BENCH_BACKEND, default nccl;
    #     gloo lets two ranks share one GPU on a test box).
    gather_mode = os.environ.get("SARA_BENCH_GATHER", "rccl")
    backend = os.environ.get("SARA_BENCH_BACKEND", "nccl")
    if world > ndev:
        local_rank = local_rank % ndev  # test boxes: ranks share devices
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    B, W, H = args.frames_per_gpu, args.width, args.height
    params = sara_amd.ImagePyramidParams(0, 6, num_octaves_max=args.octaves)
    ctx = sara_amd.SiftContext(W, H, B, params, device=local_rank)

    comm = None
    count_group = None
    native_failed = False
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if gather_mode == "rccl":
            # (gloo only ships the communicator id, the checksums and the timing
            # reduction; a bounded timeout, so that a rank that dies in one of
            # the secondary measurements cannot hang the others for half an hour)
            import datetime
            dist.init_process_group("gloo", rank=rank, world_size=world,
                                    timeout=datetime.timedelta(seconds=240))
            from sara_amd.distributed import Comm
            ident = [Comm.unique_id() if rank == 0 else None]
            dist.broadcast_object_list(ident, src=0)
            ok = 1
            try:
                comm = Comm(ctx, ident[0], world, rank, device=local_rank)
            except Exception as e:  # e.g. two ranks on one GPU of a test box
                ok = 0
                print("rank %d: native RCCL gather unavailable (%s)" % (rank, e),
                      file=sys.stderr)
            flag = torch.tensor([ok])
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0:
                if comm is not None:
                    comm.close()
                comm = None
                if world <= ndev:
                    # A node with a device per rank on which the library cannot
                    # form its communicator: measure with torch.distributed's
                    # RCCL instead of returning nothing, and say so - loudly on
                    # stderr, and in the line (config.parallelism names the
                    # transport, config.native_rccl_gather is false).
                    if os.environ.get("SARA_BENCH_STRICT_RCCL"):
                        raise SystemExit(
                            "rank %d: the RCCL communicator could not be created "
                            "although the box has %d devices for %d ranks" %
                            (rank, ndev, world))
                    native_failed = True
                    dist.destroy_process_group()
                    gather_mode, backend = "torch", "nccl"
                    dist.init_process_group("nccl", rank=rank, world_size=world,
                                            device_id=dev)
                    try:
                        count_group = dist.new_group(backend="gloo")
                    except Exception:
                        count_group = None
                    dist.all_reduce(torch.zeros(1, device=dev))
                    torch.cuda.synchronize()
                    if rank == 0:
                        print("bench: THE LIBRARY'S RCCL COMMUNICATOR COULD NOT BE "
                              "CREATED on a box with %d devices for %d ranks - the "
                              "gather runs over torch.distributed/NCCL (said so in "
                              "the line)" % (ndev, world), file=sys.stderr)
                else:
                    gather_mode, backend = "torch", "gloo"
                    if rank == 0:
                        print("bench: %d ranks on %d device(s): ranks share devices, "
                              "RCCL refuses that - the gather runs over "
                              "torch.distributed/GLOO (said so in the line)" %
                              (world, ndev), file=sys.stderr)
        elif backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=dev)
            # keypoint counts travel through host tensors on a gloo group
            try:
                count_group = dist.new_group(backend="gloo")
            except Exception:
                count_group = None
            dist.all_reduce(torch.zeros(1, device=dev))
            torch.cuda.synchronize()
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    frames_host = synth_batch(W, H, B, first_index=rank * B,
                              unique=args.unique_frames or None)
    frames = torch.from_numpy(frames_host).to(dev)  # resident in HBM
    torch.cuda.synchronize()

    legacy = world > 1 and comm is None
    # legacy path: an explicit (non-default) torch stream - the handle of
    # torch's default stream is NULL, which detect() reads as "use the
    # context's own stream", and the transfers torch posts would then not be
    # ordered after the device-to-device copies of fetch()
    stream_handle = None
    if legacy:
        bench_stream = torch.cuda.Stream(device=dev)
        torch.cuda.set_stream(bench_stream)
        stream_handle = bench_stream.cuda_stream
    feat_buf = desc_buf = so_buf = None
    pending = [None]   # legacy: point-to-point transfers in flight
    staged = [None]    # legacy: results staged on the device, not posted yet
    inflight = [None]  # native: ticket whose gather has not been done yet
    gathered = [0]     # native: keypoints that reached rank 0 in the timed region

    def step():
        """One pass over the batch; returns this rank's keypoint count."""
        if comm is not None:
            # batch i + 1 is enqueued, then batch i is gathered while it runs
            t = ctx.submit_raw(frames.data_ptr(), 0, B, W, H, on_device=True,
                               last_stage=args.stage)
            n = 0
            if inflight[0] is not None:
                res = comm.gather(inflight[0], root=0)
                n = res.counts[rank]
                gathered[0] += res.total if rank == 0 else 0
            inflight[0] = t
            return n
        ctx.detect_device(frames.data_ptr(), B, W, H, last_stage=args.stage,
                          stream=stream_handle)
        if legacy and staged[0] is not None:
            post_gather(staged[0])
            staged[0] = None
        if args.stage < 4:
            c, _, _ = ctx.extrema() if args.stage >= 2 else (np.zeros(B), 0, 0)
            return int(np.sum(c))
        counts, total = ctx.counts()
        if legacy:
            staged[0] = stage_results(total)
        return total

    def stage_results(total):
        """Legacy: OERegion[ ] (48 B), descriptors (512 B) and (s,o) pairs of
        this step copied out of the context's buffers (which the next detect()
        reuses)."""
        mine_f = torch.empty((total, 48), dtype=torch.uint8, device=dev)
        mine_d = torch.empty((total, 128), dtype=torch.float32, device=dev)
        mine_s = torch.empty((total, 2), dtype=torch.int32, device=dev)
        if total:
            capi.check(capi.load().sara_hip_sift_fetch(
                ctx._h, mine_f.data_ptr(), mine_d.data_ptr(), mine_s.data_ptr(),
                1))  # asynchronous, on the detect stream
        return [mine_f, mine_d, mine_s], total

    def post_gather(item):
        """Legacy gatherv to rank 0 (sara_amd/distributed.py)."""
        tensors, total = item
        if backend != "nccl":
            tensors = [t.cpu() for t in tensors]  # gloo: host staging
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
        """Drains the pipeline; returns this rank's keypoints of the drained
        batch (native path)."""
        n = 0
        if comm is not None and inflight[0] is not None:
            res = comm.gather(inflight[0], root=0)
            n = res.counts[rank]
            gathered[0] += res.total if rank == 0 else 0
            inflight[0] = None
        if legacy:
            if staged[0] is not None:
                post_gather(staged[0])
                staged[0] = None
            wait_gather()
        torch.cuda.synchronize()
        ctx.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()
        return n

    for _ in range(args.warmup):
        step()
    sync()
    gathered[0] = 0
    stage_ms = {}
    kp_local = 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        kp_local += step()
        if args.stage >= 1:
            for k, v in ctx.stage_times().items():
                stage_ms[k] = stage_ms.get(k, 0.0) + v
    kp_local += sync()
    elapsed = time.perf_counter() - t0

    on_dev = world > 1 and legacy and backend == "nccl"
    rdev = dev if on_dev else torch.device("cpu")
    el = torch.tensor([elapsed], device=rdev, dtype=torch.float64)
    kp = torch.tensor([kp_local], device=rdev, dtype=torch.int64)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(kp, op=dist.ReduceOp.SUM)
    elapsed = float(el.item())
    kp_total = int(kp.item())
    if comm is not None and rank == 0 and gathered[0] != kp_total:
        raise SystemExit("gather: %d keypoints reached rank 0, the ranks "
                         "produced %d" % (gathered[0], kp_total))
    gather_check = h2h_multi = h2h_multi_error = None
    if comm is not None and args.stage >= 5:
        gather_check = verify_gather(ctx, comm, frames, args, dist, rank, world)
    if world > 1 and args.stage >= 5:
        if not args.no_extras:
            # a secondary measurement: whatever goes wrong in it (shared memory,
            # host registration across processes - untested on a real node) must
            # not cost the line
            try:
                h2h_multi = host_to_host_multi(
                    ctx, frames_host, args, torch, dist, rank, world,
                    kp_total / max(args.steps, 1) / world)
            except Exception as e:  # noqa: BLE001
                h2h_multi = None
                h2h_multi_error = repr(e)
                sys.stderr.write("rank %d: host_to_host_multi failed: %r\n" % (rank, e))
                if args.strict_h2h:
                    raise SystemExit("host_to_host_multi failed and --strict-h2h "
                                     "is set: %r" % (e,))

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
            "value_definition": (
                "HBM-resident: frames in HBM when the timed region starts, "
                "keypoint arrays in HBM%s when it ends.  SURVEY.md 8d's metric "
                "(pinned host frames -> results in host memory) is "
                "value_host_to_host_*" % (" of rank 0" if world > 1 else "")),
            "config": {
                "workload": "full SIFT (pyramid+DoG, extrema+refine, polar "
                            "gradients, orientations, 128-D descriptors) on "
                            "%d synthetic %dx%d frames per GPU, %d octaves x 3 "
                            "scales/octave, frames resident in HBM" %
                            (B, W, H, args.octaves),
                "frames_per_gpu": B,
                "global_frames": B * world,
                "distinct_frames_per_gpu": args.unique_frames or B,
                "frame_seeds": "1234 + global frame index (SplitMix64, "
                               "include/sara_synth.h)",
                "keypoints_per_frame": kp_total / (steps * B * world),
                "frames_per_s": steps * B * world / elapsed,
                "parallelism": ("frames sharded %d/GPU, gatherv of keypoints to "
                                "rank 0: %s" % (B, "library RCCL (ncclAllGather "
                                "of counts + grouped ncclSend/ncclRecv), "
                                "pipelined under the next batch"
                                if comm is not None else
                                "torch.distributed/" + backend))
                               if world > 1 else "single GPU",
                "last_stage": args.stage,
                "native_rccl_gather": (comm is not None) if world > 1 else None,
                "native_rccl_gather_failed": bool(native_failed),
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
                "traffic_source": (
                    "profiles/pyramid_traffic.json: rocprofv3 --pmc FETCH_SIZE / "
                    "WRITE_SIZE passes of an earlier run of this command (x2 per "
                    "the guide, calibrated by tools/ubench/fetch_calib.hip); a "
                    "static file, NOT measured in this run") if traffic else None,
                "launches_per_step": pyr_launches,
                "avg_launch_us": 1e3 * pyr_ms / max(pyr_launches, 1),
                "algorithmic_bytes_per_step": bytes_frame * B,
                "us_per_frame": 1e3 * pyr_ms / B,
            },
        }
        if world > 1:
            out["transport"] = (comm.transport if comm is not None else
                                "torch.distributed/" + backend)
            out["rccl_nranks"] = world if out["transport"] == "rccl" else 0
            out["rccl_version"] = (rccl_version_string()
                                   if out["transport"] == "rccl" else None)
            out["gather_verified"] = gather_check is not None
            out["launch"] = "one process per GPU (torch.distributed.run)"
        if gather_check is not None:
            out["config"]["gather_verified"] = gather_check
        if h2h_multi is not None:
            out["value_host_to_host_gray8"] = h2h_multi["keypoints_per_s"]
            out["value_8d_gray8"] = h2h_multi["keypoints_per_s"]
            out["config"]["host_to_host"] = h2h_multi
        elif world > 1 and args.stage >= 5 and not args.no_extras:
            # SURVEY 8d's figure is part of an N > 1 line: when it could not be
            # measured the key is there, null, with the reason next to it
            # (--strict-h2h turns that into a failed run)
            out["value_host_to_host_gray8"] = None
            out["value_8d_gray8"] = None
            out["host_to_host_error"] = h2h_multi_error
        if world == 1 and not args.no_extras:
            # the same stage on ONE stream, an event pair around every launch
            rows, serial_ms = pyramid_per_kernel(args, torch, dev, frames)
            out["roofline"]["serial_ms"] = serial_ms
            out["roofline"]["serial_frac"] = (
                bytes_frame * B / 1e9 / (serial_ms / 1e3) / HBM_PEAK_GBS)
            out["roofline"]["frac_definition"] = (
                "frac: per-octave streams overlapped (the shipped schedule), "
                "first launch -> all octaves joined; serial_frac: the same "
                "launches on one stream, sum of their durations")
            out["roofline"]["per_kernel"] = rows
            worst = min((r for r in rows if r["radius"] > 0), key=lambda r: r["frac"])
            out["roofline"]["worst_kernel"] = worst
        if world == 1 and args.cpu_frames > 0:
            oracle_results, out["cpu_baseline"] = cpu_baseline(args, frames_host)
            out["config"]["gpu_over_cpu"] = out["value"] / max(
                out["cpu_baseline"]["value"], 1e-9)
            if args.stage >= 5:
                # the benchmarked batch itself (same launch geometry) against
                # the oracle frames just computed - outside the timed region
                ctx.detect_device(frames.data_ptr(), B, W, H, last_stage=5)
                n_ok, worst = check_parity(ctx.fetch(), oracle_results)
                out["parity_checked_frames"] = n_ok
                out["config"]["parity"] = (
                    "frames 0..%d of the benchmarked %d-frame batch equal the "
                    "CPU oracle: counts, order, (s,o), coordinates exact, "
                    "orientation 1e-6 rad, descriptors max-abs %.2g (bar 2e-3)"
                    % (n_ok - 1, B, worst))
        if world == 1 and not args.no_extras and args.stage >= 5:
            h2h = host_to_host(ctx, frames_host, args, torch)
            out["value_host_to_host_gray8"] = h2h["gray8"]["keypoints_per_s"]
            out["value_host_to_host_float32"] = h2h["float32"]["keypoints_per_s"]
            # SURVEY.md 8d's own metric under the names the verdict reads:
            # pinned host frames in -> keypoints + descriptors in host memory
            out["value_8d_float32"] = h2h["float32"]["keypoints_per_s"]
            out["value_8d_gray8"] = h2h["gray8"]["keypoints_per_s"]
            out["config"]["workload"] += (
                "; `value` = frames and results resident in HBM; SURVEY 8d "
                "host->host: %.1f M kp/s float32, %.1f M gray8"
                % (h2h["float32"]["keypoints_per_s"] / 1e6,
                   h2h["gray8"]["keypoints_per_s"] / 1e6))
            out["config"]["host_to_host_keypoints_per_s"] = \
                h2h["gray8"]["keypoints_per_s"]
            out["config"]["host_to_host"] = {
                "definition": "SURVEY.md 8d: frames in pinned host memory -> "
                              "OERegion[] + descriptors in pinned host memory, "
                              "two batches in flight (submit/collect); `value` "
                              "above is the HBM-resident rate",
                "gray8": h2h["gray8"], "float32": h2h["float32"],
                "hip_runtime": [l.split()[-1] for l in open("/proc/self/maps")
                                if "libamdhip64" in l][:1]}
            ctx.close()
            sysrt = host_to_host_system_runtime(args)
            if sysrt is not None:
                # the library as a C++ caller loads it: the image's ROCm runtime
                # instead of the one inside the torch wheel
                out["config"]["host_to_host"]["without_torch_in_the_process"] = sysrt
                out["value_host_to_host_float32_system_runtime"] = \
                    sysrt["float32"]["keypoints_per_s"]
            out["config"].update(secondary_configs(args, torch, dev))
            out["config"]["width_not_multiple_of_4"] = odd_width_case(torch, dev)
            out["config"]["match"] = match_config(torch, dev)
        print(json.dumps(out))
    if comm is not None:
        comm.close()
    ctx.close()
    if world > 1:
        try:
            dist.barrier()
            dist.destroy_process_group()
        except Exception as e:  # noqa: BLE001 - the line is out already
            sys.stderr.write("rank %d: shutdown: %r\n" % (rank, e))


if __name__ == "__main__":
    main()
