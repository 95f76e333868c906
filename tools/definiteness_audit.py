"""Prints the audit DESIGN.md section 2 quotes: the oracle's three definiteness
rules on every Hessian refine_extremum examines, and the descriptor delta
between Eigen's packet order and the left-to-right squaredNorm(), over the
golden photograph and N synthetic 1080p benchmark frames (default 64).
CPU only:  python tools/definiteness_audit.py [N]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import refbind as rb  # noqa: E402
import common  # noqa: E402
from sara_amd.synth import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
rb.build()
params = rb.PyramidParams(0, 6, None, 1, 0.5, 1.6, 4)
images = [common.load_sunflower_gray()] + [synth(1920, 1080, 1234 + i)
                                           for i in range(n)]
for mode, name in ((0, "default build"), (1, "signed extremum type")):
    with rb.detector_mode(mode), rb.definiteness_audit() as a:
        for img in images:
            rb.RefSift(img, params, parallel=True).keypoints()
        print(name, a.read())
mx, diff, tot = 0.0, 0, 0
for img in images[:9]:
    d0 = rb.RefSift(img, params, parallel=True).keypoints()[2]
    with rb.squared_norm_order(1):
        d1 = rb.RefSift(img, params, parallel=True).keypoints()[2]
    mx = max(mx, float(np.abs(d0 - d1).max()))
    diff += int(np.count_nonzero(d0 != d1))
    tot += d0.size
print("normalize(): max |delta| %.3g on 0..255, %d of %d bins differ" % (mx, diff, tot))
