"""One-off: an 8K frame (7680 x 4320, uncapped octave count) against the oracle."""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import conftest, refbind as rb
import test_gpu_pipeline as T
import sara_amd
from sara_amd.synth import synth
rb.build(); rb.lib().ref_omp_set_threads(conftest._usable_cpus())
w, h = 7680, 4320
img = synth(w, h, 77)
t = time.time()
ref = rb.RefSift(img, T.ref_params(rb, 0, 2**31 - 1), parallel=True)
print("oracle s", time.time() - t, "octaves", ref.octave_count)
with sara_amd.SiftContext(w, h, 1, T.hip_params(0, 2**31 - 1)) as ctx:
    ctx.detect(img)
    T.compare_full(ctx, ref)
    print(T.compare_lists(T.run_lists(ctx), ref, 0))
print("8K ok")
