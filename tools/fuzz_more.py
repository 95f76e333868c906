"""One-off wider fuzz of the GPU/oracle parity (same checks as
tests/test_gpu_pipeline.py::test_fuzz_parity, more cases, other seeds, plus
batches, gauss_truncate, 8-bit input and the matcher).
  python tools/fuzz_more.py [n_cases] [seed] [size_factor]"""
import os, sys
os.environ.setdefault("SARA_HIP_MARCH_MIN_PIXELS", "0")
os.environ.setdefault("SARA_HIP_STRIP_GROUP", "8")
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import conftest, refbind as rb
import test_gpu_pipeline as T
import test_gpu_matching as M
import sara_amd
from sara_amd.synth import synth

rb.build()
rb.lib().ref_omp_set_threads(conftest._usable_cpus())
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 777
size_factor = int(sys.argv[3]) if len(sys.argv) > 3 else 1
rng = np.random.default_rng(seed + 1)
bad = 0
for case in T._fuzz_cases(n, seed):
    i, w, h, first, scales, kfac, cam, noct, thres, edge, iters = case
    w, h = w * size_factor, h * size_factor
    if size_factor > 1 and first < 0:
        first = 0
    try:
        batch = int(rng.choice([1, 1, 2, 5]))
        trunc = float(rng.choice([4.0, 4.0, 3.0, 2.5]))
        imgs = np.stack([synth(w, h, 9000 + 7 * i + seed + b) for b in range(batch)])
        use_u8 = bool(rng.integers(0, 4) == 0)
        if use_u8:
            u8 = np.clip(imgs * 255.0, 0, 255).astype(np.uint8)
            imgs = u8.astype(np.float32) / np.float32(255)
        # "corrected" mode switches and RootSIFT on a third of the cases
        mode = int(rng.choice([0, 0, 0, 0, 1, 2, 3]))
        root = bool(rng.integers(0, 3) == 0)
        import math
        if (mode & 2) and round(math.log(2.0) / math.log(kfac if kfac else 2 ** (1 / 3))) >= scales:
            mode &= 1
        with sara_amd.SiftContext(w, h, batch, T.hip_params(first, noct, cam, scales, kfac),
                                  gauss_truncate=trunc, extremum_thres=thres,
                                  edge_ratio_thres=edge,
                                  extremum_refinement_iter=iters) as ctx:
            ctx.set_option(sara_amd.capi.OPT_SIGNED_EXTREMUM_TYPE, mode & 1)
            ctx.set_option(sara_amd.capi.OPT_DOWNSCALE_AT_DOUBLE_SIGMA, (mode >> 1) & 1)
            if use_u8:
                ctx.detect_u8(u8)
            else:
                ctx.detect(imgs)
            lists = T.run_lists(ctx)
            descs = []
            for b in range(batch):
                with rb.detector_mode(mode):
                    ref = rb.RefSift(imgs[b], T.ref_params(rb, first, noct, cam, scales, kfac),
                                     gauss_truncate=trunc, extremum_thres=thres,
                                     edge_ratio_thres=edge, extremum_refinement_iter=iters)
                T.compare_full(ctx, ref, frame=b)
                T.compare_lists(lists, ref, b)
                descs.append(ref.keypoints()[2])
            if root and int(lists[3].sum()):
                ctx.set_option(sara_amd.capi.OPT_ROOT_SIFT, 1)
                (ctx.detect_u8(u8) if use_u8 else ctx.detect(imgs))
                got = ctx.fetch()[2]
                want = rb.root_sift(lists[5])
                assert np.allclose(got, want, rtol=1e-5, atol=1e-7), "root sift"
                ctx.set_option(sara_amd.capi.OPT_ROOT_SIFT, 0)
                (ctx.detect_u8(u8) if use_u8 else ctx.detect(imgs))
            if batch >= 2 and len(descs[0]) >= 2 and len(descs[1]) >= 2:
                ratio = float(rng.choice([0.6, 0.8, 1.0]))
                M.assert_same(ctx.match_frames(0, 1, ratio),
                              rb.compute_matches(ctx.keypoint_lists()[0].descriptor_matrix,
                                                 ctx.keypoint_lists()[1].descriptor_matrix, ratio))
    except Exception as e:  # noqa
        bad += 1
        import traceback
        tb = traceback.extract_tb(e.__traceback__)
        where = "; ".join("%s:%d %s" % (os.path.basename(f.filename), f.lineno, f.name) for f in tb[-3:])
        print("FAIL", case, "batch", batch, "u8", use_u8, "mode", mode, "root", root,
              repr(e)[:300], "|", where, flush=True)
print("cases", n, "failures", bad)
