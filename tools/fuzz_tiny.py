"""One-off fuzz of tiny and extreme-aspect images (2..48 pixels a side, 2-8 x 300-2000 strips):
GPU against the oracle; inputs for which the reference would build no octave (it then dereferences
an empty pyramid, ImagePyramid.hpp:292-295) must come back empty.  python tools/fuzz_tiny.py"""
import os, sys, faulthandler
faulthandler.enable()
os.environ.setdefault("SARA_HIP_MARCH_MIN_PIXELS", "0")
os.environ.setdefault("SARA_HIP_STRIP_GROUP", "8")
root = os.getcwd()
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import conftest, refbind as rb
import test_gpu_pipeline as T
import sara_amd
from sara_amd.synth import synth
rb.build(); rb.lib().ref_omp_set_threads(conftest._usable_cpus())
rng = np.random.default_rng(3)
bad = 0
for i in range(300):
    w = int(rng.integers(2, 49)); h = int(rng.integers(2, 49))
    if i % 10 == 0: w, h = int(rng.integers(2, 9)), int(rng.integers(300, 900))
    if i % 10 == 1: w, h = int(rng.integers(300, 2000)), int(rng.integers(2, 9))
    first = int(rng.choice([0, 0, -1, 1]))
    noct = int(rng.integers(1, 8))
    try:
        img = synth(w, h, 100 + i)
        # octaves the reference would build (GaussianPyramid.hpp:80-94)
        p = T.hip_params(first, noct)
        nref = p.octave_count(w, h)
        with sara_amd.SiftContext(w, h, 1, p) as ctx:
            ctx.detect(img)
            if nref <= 0:
                # the reference dereferences an empty pyramid here (UB)
                assert ctx.fetch()[0].sum() == 0
                continue
            ref = rb.RefSift(img, T.ref_params(rb, first, noct))
            T.compare_full(ctx, ref)
            T.compare_lists(T.run_lists(ctx), ref, 0)
    except Exception as e:
        bad += 1
        print("FAIL", (w, h, first, noct), repr(e)[:200], flush=True)
print("tiny cases 300 failures", bad)
