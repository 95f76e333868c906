"""One-off wider fuzz of the GPU/oracle parity (same checks as
tests/test_gpu_pipeline.py::test_fuzz_parity, more cases, other seeds).
  python tools/fuzz_more.py [n_cases] [seed]"""
import os, sys
os.environ.setdefault("SARA_HIP_MARCH_MIN_PIXELS", "0")
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import conftest, refbind as rb
import test_gpu_pipeline as T
import sara_amd
from sara_amd.synth import synth

rb.build()
rb.lib().ref_omp_set_threads(conftest._usable_cpus())
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 777
bad = 0
for case in T._fuzz_cases(n, seed):
    i, w, h, first, scales, kfac, cam, noct, thres, edge, iters = case
    try:
        img = synth(w, h, 9000 + i + seed)
        ref = rb.RefSift(img, T.ref_params(rb, first, noct, cam, scales, kfac),
                         extremum_thres=thres, edge_ratio_thres=edge,
                         extremum_refinement_iter=iters)
        with sara_amd.SiftContext(w, h, 1, T.hip_params(first, noct, cam, scales, kfac),
                                  extremum_thres=thres, edge_ratio_thres=edge,
                                  extremum_refinement_iter=iters) as ctx:
            ctx.detect(img)
            T.compare_full(ctx, ref)
            T.compare_lists(T.run_lists(ctx), ref, 0)
    except Exception as e:  # noqa
        bad += 1
        print("FAIL", case, repr(e)[:300], flush=True)
print("cases", n, "failures", bad)
