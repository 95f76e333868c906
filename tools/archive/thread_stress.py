"""Stress: several host threads, each with its own contexts, creating /
destroying / capturing / running concurrently (what the per-thread context cache
of compute_sift_keypoints and the one-thread-per-rank group do)."""
import sys
import threading

import numpy as np

import sara_amd
from sara_amd.synth import synth

p = sara_amd.ImagePyramidParams(0, 6, num_octaves_max=3)
errors = []


def worker(k):
    try:
        for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
            for w in (200, 208, 216, 224, 232, 240):
                sara_amd.compute_sift_keypoints(synth(w, 160, 3), p)
    except Exception as e:  # noqa: BLE001
        errors.append(e)


ts = [threading.Thread(target=worker, args=(k,)) for k in range(6)]
for t in ts:
    t.start()
for t in ts:
    t.join()
print("errors:", errors)
