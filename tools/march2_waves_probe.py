"""Developer probe (verdict r05 item 5): do the R >= 8 blurs want more resident
waves?  The hand-scheduled 2-column kernel needs 86 / 100 / 114 VGPRs (R = 8 /
10 / 12): 5 / 4 / 4 waves per SIMD fit, and how many are resident is set by the
launch size (SARA_HIP_OPT_MARCH_WAVES: target waves per launch; a launch rounds
up to whole segments).  64 x 1080p, octave 0's three wide blurs, single stream
(kernel durations) and the whole pyramid stage with the per-octave streams."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import sara_amd  # noqa: E402
from sara_amd import capi  # noqa: E402
from sara_amd.synth import synth_batch  # noqa: E402

B, W, H = 64, 1920, 1080
frames = torch.from_numpy(synth_batch(W, H, B, unique=8)).cuda()
p = sara_amd.ImagePyramidParams(0, 6, num_octaves_max=4)
P = sum((W >> o) * (H >> o) for o in range(4))
print("march2 waves | octave 0: R=8 R=10 R=12 us | octave 1: R=8 R=10 R=12 | "
      "octave 2: R=8 R=10 R=12 | serial sum us | pyramid stage ms (streams) frac")
with sara_amd.SiftContext(W, H, B, p) as ctx:
    ctx.set_option(capi.OPT_KERNEL_SELECTION, capi.SELECT_SHIPPED)
    for waves in (2048, 2560, 3072, 3584, 4096, 5120):
        ctx.set_option(capi.OPT_MARCH2_WAVES, waves)
        ctx.set_option(capi.OPT_SINGLE_STREAM, 1)
        ctx.set_option(capi.OPT_LAUNCH_TIMERS, 1)
        us = {(o, t): [] for o in range(3) for t in (17, 21, 25)}
        total = []
        for it in range(6):
            ctx.detect_device(frames.data_ptr(), B, W, H, last_stage=1)
            ctx.synchronize()
            if it >= 2:
                rows = ctx.pyramid_launches()
                total.append(1e3 * float(rows["ms"].sum()))
                for r in rows:
                    if (int(r["octave"]), int(r["taps"])) in us:
                        us[(int(r["octave"]), int(r["taps"]))].append(1e3 * float(r["ms"]))
        ctx.set_option(capi.OPT_LAUNCH_TIMERS, 0)
        ctx.set_option(capi.OPT_SINGLE_STREAM, 0)
        pyr = []
        for it in range(8):
            ctx.detect_device(frames.data_ptr(), B, W, H, last_stage=1)
            ctx.synchronize()
            if it >= 2:
                pyr.append(ctx.stage_times()["pyramid"])
        m = {k: float(np.mean(v)) if v else 0.0 for k, v in us.items()}
        print("%8d | %7.1f %6.1f %6.1f | %6.1f %6.1f %6.1f | %6.1f %6.1f %6.1f | %8.1f | %8.3f %6.3f" % (
            waves, m[(0, 17)], m[(0, 21)], m[(0, 25)], m[(1, 17)], m[(1, 21)],
            m[(1, 25)], m[(2, 17)], m[(2, 21)], m[(2, 25)], float(np.mean(total)),
            float(np.mean(pyr)),
            48.0 * P * B / 1e9 / (np.mean(pyr) / 1e3) / 8000.0), flush=True)
