"""Developer probe: per-launch device times of the pyramid
blurs of the 64 x 1080p batch on ONE stream, exact arithmetic against the
hand-written FMA form of the same kernels (SARA_HIP_OPT_FMA_BLUR) - 88 against
50 arithmetic instructions per pixel at R = 12, same registers, same
occupancy.  If the time follows the instruction count the exact kernels are
issue-bound."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import sara_amd  # noqa: E402
from sara_amd import capi  # noqa: E402
from sara_amd.synth import synth_batch  # noqa: E402

B, W, H = (int(sys.argv[1]) if len(sys.argv) > 1 else 64), 1920, 1080
frames = torch.from_numpy(synth_batch(W, H, B, unique=min(B, 8))).to("cuda:0")
params = sara_amd.ImagePyramidParams(0, 6, num_octaves_max=4)
res = {}
with sara_amd.SiftContext(W, H, B, params) as c:
    c.set_option(capi.OPT_SINGLE_STREAM, 1)
    c.set_option(capi.OPT_GRAPH_REPLAY, 0)
    c.set_option(capi.OPT_LAUNCH_TIMERS, 1)
    for name, fma in (("exact", 0), ("fma", 1)):
        c.set_option(capi.OPT_FMA_BLUR, fma)
        acc, reps = {}, 0
        for i in range(7):
            c.detect_device(frames.data_ptr(), B, W, H, last_stage=1)
            c.synchronize()
            if i < 2:
                continue
            reps += 1
            for r in c.pyramid_launches():
                k = (int(r["octave"]), int(r["scale"]), int(r["taps"]), int(r["pixels"]))
                acc[k] = acc.get(k, 0.0) + float(r["ms"])
        res[name] = {k: v / reps for k, v in acc.items()}
print("octave scale radius   exact us    fma us   fma/exact   exact frac of 8 TB/s")
for k in sorted(res["exact"]):
    o, s, taps, px = k
    e, f = res["exact"][k], res["fma"].get(k, float("nan"))
    print("%5d %5d %6d %10.1f %9.1f %10.2f %12.3f"
          % (o, s, taps // 2, 1e3 * e, 1e3 * f, f / e, 8 * px / 1e9 / (e / 1e3) / 8000))
print("serial stage: exact %.3f ms, fma %.3f ms"
      % (sum(res["exact"].values()), sum(res["fma"].values())))
