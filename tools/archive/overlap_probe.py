#!/usr/bin/env python3
"""How much of a 64 x 1080p step overlaps with its neighbours?

The step is 4.3 ms of bandwidth-bound streaming kernels followed by 2.4 ms of
instruction-issue-bound per-keypoint kernels.  Four schedules of the same
work, frames and results resident in HBM:

  serial     detect_device(); counts()            (what bench.py times)
  pipelined  submit(i + 1); ticket_counts(i)      one context, two result slots
  two-ctx    two contexts with their own pyramids, alternating: the kernels
             of batch i + 1 may run next to those of batch i
  halves     two contexts of 32 frames, one step = one submit on each

Prints ms per 64-frame step for each.
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch  # noqa: E402  (device memory only)

import sara_amd  # noqa: E402
from sara_amd.synth import synth_batch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    args = ap.parse_args()
    B, W, H = args.frames, 1920, 1080
    frames = torch.from_numpy(synth_batch(W, H, B)).cuda()
    torch.cuda.synchronize()

    def make(batch):
        return sara_amd.SiftContext(
            W, H, batch, sara_amd.ImagePyramidParams(0, 6, num_octaves_max=4))

    def drain(ctx, t):
        _, total = ctx.ticket_counts(t)
        ctx.collect_into(t, None, None, None)
        return total

    def run(name, fn, sync):
        for _ in range(args.warmup):
            fn()
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            fn()
        sync()
        dt = time.perf_counter() - t0
        print("%-10s %.3f ms per %d-frame step" % (name, 1e3 * dt / args.steps, B),
              flush=True)

    # serial
    ctx = make(B)

    def serial():
        ctx.detect_device(frames.data_ptr(), B, W, H)
        ctx.counts()
    run("serial", serial, ctx.synchronize)

    # pipelined, one context
    pend = []

    def pipelined():
        pend.append(ctx.submit_raw(frames.data_ptr(), 0, B, W, H, on_device=True))
        if len(pend) > 1:
            drain(ctx, pend.pop(0))

    def pipe_sync():
        while pend:
            drain(ctx, pend.pop(0))
    run("pipelined", pipelined, pipe_sync)

    # two contexts alternating
    ctx2 = make(B)
    ring = []
    turn = [0]

    def two_ctx():
        c = (ctx, ctx2)[turn[0] & 1]
        turn[0] += 1
        ring.append((c, c.submit_raw(frames.data_ptr(), 0, B, W, H, on_device=True)))
        if len(ring) > 2:
            drain(*ring.pop(0))

    def ring_sync():
        while ring:
            drain(*ring.pop(0))
    run("two-ctx", two_ctx, ring_sync)
    del ctx2

    # halves
    h = B // 2
    ca, cb = make(h), make(h)
    second = frames.data_ptr() + h * W * H * 4

    def halves():
        ring.append((ca, ca.submit_raw(frames.data_ptr(), 0, h, W, H, on_device=True)))
        ring.append((cb, cb.submit_raw(second, 0, h, W, H, on_device=True)))
        while len(ring) > 2:
            drain(*ring.pop(0))
    run("halves", halves, ring_sync)


if __name__ == "__main__":
    main()
