"""Developer probe (run on the GPU box through gpurun): stage-by-stage parity
of the HIP path against the CPU oracle on one synthetic frame, plus stage
timings for a batch.  Prints; asserts nothing."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np

import sara_amd
from sara_amd.synth import synth, synth_batch
import refbind as rb


def main():
    w = int(sys.argv[1]) if len(sys.argv) > 1 else 640
    h = int(sys.argv[2]) if len(sys.argv) > 2 else 480
    batch = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    noct = 4
    img = synth(w, h, 1234)
    hp = sara_amd.ImagePyramidParams(0, num_octaves_max=noct)
    rp = rb.PyramidParams(0, 6, None, 1, 0.5, 1.6, noct)
    t = time.time()
    ref = rb.RefSift(img, rp, parallel=True)
    print("oracle %.2fs" % (time.time() - t), ref.times())
    ctx = sara_amd.SiftContext(w, h, batch, hp)
    ctx.set_option(sara_amd.capi.OPT_ALL_GRADIENT_SCALES, 1)
    ctx.detect(img)
    print("octaves", ctx.octave_count, ref.octave_count)
    for o in range(ctx.octave_count):
        for s in range(6):
            g, r = ctx.gaussian(s, o), ref.gaussian(s, o)
            print("G(%d,%d) exact=%s maxabs=%.3g" % (s, o, np.array_equal(g, r),
                                                    np.abs(g - r).max()))
        for s in range(5):
            g, r = ctx.dog(s, o), ref.dog(s, o)
            print("D(%d,%d) exact=%s maxabs=%.3g" % (s, o, np.array_equal(g, r),
                                                    np.abs(g - r).max()))
        for s in range(6):
            g, r = ctx.gradient(s, o), ref.gradient(s, o)
            print("grad(%d,%d) mag exact=%s ori exact=%s maxabs=%.3g nbad=%d" % (
                s, o, np.array_equal(g[..., 0], r[..., 0]),
                np.array_equal(g[..., 1], r[..., 1]), np.abs(g - r).max(),
                int((g != r).sum())))
    c, reg, xyso = ctx.extrema()
    rreg, rxyso = ref.extrema()
    print("extrema", c, len(rreg))
    if len(reg) == len(rreg):
        print(" site equal", np.array_equal(xyso, rxyso))
        for f in ("coords", "shape_matrix", "extremum_value", "extremum_type"):
            eq = np.array_equal(reg[f], rreg[f])
            d = np.abs(reg[f].astype(np.float64) - rreg[f].astype(np.float64)).max()
            print("  %s exact=%s maxabs=%.3g nbad=%d" % (
                f, eq, d, int((reg[f] != rreg[f]).sum())))
    else:
        a = set(map(tuple, xyso.tolist()))
        b = set(map(tuple, rxyso.tolist()))
        print(" only gpu", sorted(a - b)[:10], "only ref", sorted(b - a)[:10])
    kc, kreg, kdesc, kso = ctx.fetch()
    rk, rso, rdesc = ref.keypoints()
    print("keypoints", kc, len(rk))
    if len(kreg) == len(rk):
        print(" so equal", np.array_equal(kso, rso))
        for f in ("coords", "shape_matrix", "orientation", "extremum_value",
                  "type", "extremum_type"):
            eq = np.array_equal(kreg[f], rk[f])
            d = np.abs(kreg[f].astype(np.float64) - rk[f].astype(np.float64)).max()
            print("  %s exact=%s maxabs=%.3g nbad=%d" % (
                f, eq, d, int((kreg[f] != rk[f]).sum())))
        dd = np.abs(kdesc - rdesc)
        print("  desc maxabs=%.4g mean=%.3g nrows>1e-2: %d" % (
            dd.max(), dd.mean(), int((dd.max(axis=1) > 1e-2).sum())))
        worst = int(dd.max(axis=1).argmax())
        print("  worst row", worst, kreg[worst], kso[worst])
    print("stage ms (1 frame)", ctx.stage_times())

    frames = synth_batch(w, h, batch, unique=min(batch, 2))
    for it in range(3):
        ctx.detect(frames)
        ctx.synchronize()
        print("stage ms (batch %d)" % batch, ctx.stage_times())
    c, tot = ctx.counts()
    print("counts", c, tot)


if __name__ == "__main__":
    main()
