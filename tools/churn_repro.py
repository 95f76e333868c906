#!/usr/bin/env python3
"""Reproducer of the hip::Graph::UpdateStreams crash (ROCm 7.0 runtime as
bundled with torch): several host threads, each creating a context, running one
graph-replayed detection and destroying contexts, at the same time.

  python tools/churn_repro.py [--no-torch] [--lock] [--threads N] [--reps R]

--lock serialises every call into the library with one Python lock (an
experiment: does full serialisation avoid the crash?)."""
import argparse
import os
import sys
import threading

ap = argparse.ArgumentParser()
ap.add_argument("--no-torch", action="store_true")
ap.add_argument("--lock", action="store_true")
ap.add_argument("--threads", type=int, default=6)
ap.add_argument("--reps", type=int, default=4)
ap.add_argument("--main-busy", action="store_true",
                help="the main thread keeps replaying its own graph meanwhile")
args = ap.parse_args()
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
if not args.no_torch:
    import torch  # noqa: F401  (its libamdhip64 becomes the process's runtime)
import sara_amd  # noqa: E402
from sara_amd.synth import synth  # noqa: E402

p = sara_amd.ImagePyramidParams(0, 6, num_octaves_max=3)
big = threading.Lock()
errors = []


def worker(k):
    try:
        for rep in range(args.reps):
            for w in (200, 208, 216, 224, 232, 240):
                img = synth(w, 160, 3)
                if args.lock:
                    with big:
                        sara_amd.compute_sift_keypoints(img, p)
                else:
                    sara_amd.compute_sift_keypoints(img, p)
    except Exception as e:  # noqa: BLE001
        errors.append(e)


sara_amd.compute_sift_keypoints(synth(320, 240, 1), p)
ts = [threading.Thread(target=worker, args=(k,)) for k in range(args.threads)]
for t in ts:
    t.start()
if args.main_busy:
    frame = synth(320, 240, 1)
    with sara_amd.SiftContext(320, 240, 1, p) as mine:
        n = 0
        while any(t.is_alive() for t in ts):
            mine.detect(frame)
            mine.counts()
            n += 1
    print("main thread: %d replays meanwhile" % n)
for t in ts:
    t.join()
print("done, errors:", errors, flush=True)
