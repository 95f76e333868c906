"""Generates the committed golden fixtures (run in the build container, where
/root/reference exists; the tests never read /root/reference).

Inputs are DATA only: the reference's sample photograph
data/sunflowerField.jpg decoded to RGB8 and stored losslessly as PNG, so the
GPU box does not depend on a JPEG decoder version.  Expected outputs come from
the CPU oracle (oracle/sift_ref.hpp), which is itself pinned by the
reference's unit tests (tests/test_oracle_reference_pins.py).

  sunflower_rgb8.png          1600x1200 RGB8 (config 1 input)
  sunflower_full.npz          oracle keypoints of the full frame, 4 octaves
                              (regions, scale/octave pairs, descriptor
                              checksums; descriptors of every 8th keypoint)
  sunflower_crop.npz          512x384 crop: extrema, keypoints, descriptors,
                              and checksums of every pyramid plane
"""
import hashlib
import os
import sys

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import refbind as rb  # noqa: E402

SRC = "/root/reference/data/sunflowerField.jpg"
CROP = (544, 408, 512, 384)  # x0, y0, w, h


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    rgb = np.array(Image.open(SRC).convert("RGB"))
    Image.fromarray(rgb).save(os.path.join(HERE, "sunflower_rgb8.png"),
                              optimize=True)
    gray = rb.rgb8_to_gray32f(rgb)
    params = rb.PyramidParams(0, 6, None, 1, 0.5, 1.6, 4)

    full = rb.RefSift(gray, params, parallel=True)
    reg, so, desc = full.keypoints()
    ereg, exyso = full.extrema()
    np.savez_compressed(
        os.path.join(HERE, "sunflower_full.npz"),
        gray_sha256=sha(gray), regions=reg.view(np.uint8).reshape(-1, 48),
        scale_octave=so, extrema_xyso_type=exyso,
        desc_every8=desc[::8], desc_row_sums=desc.sum(axis=1),
        n_extrema=len(ereg), n_keypoints=len(reg))

    x0, y0, w, h = CROP
    crop = np.ascontiguousarray(gray[y0:y0 + h, x0:x0 + w])
    r = rb.RefSift(crop, params, parallel=True)
    reg, so, desc = r.keypoints()
    ereg, exyso = r.extrema()
    planes = {}
    for o in range(r.octave_count):
        for s in range(6):
            planes["G_%d_%d" % (s, o)] = sha(r.gaussian(s, o))
            planes["grad_%d_%d" % (s, o)] = sha(r.gradient(s, o))
        for s in range(5):
            planes["D_%d_%d" % (s, o)] = sha(r.dog(s, o))
    np.savez_compressed(
        os.path.join(HERE, "sunflower_crop.npz"),
        crop=np.array(CROP), crop_sha256=sha(crop),
        regions=reg.view(np.uint8).reshape(-1, 48), scale_octave=so,
        descriptors=desc, extrema=ereg.view(np.uint8).reshape(-1, 48),
        extrema_xyso_type=exyso,
        plane_names=np.array(sorted(planes)),
        plane_sha256=np.array([planes[k] for k in sorted(planes)]),
        G_3_1=r.gaussian(3, 1), D_2_2=r.dog(2, 2))
    print("full: %d extrema, %d keypoints; crop: %d extrema, %d keypoints" %
          (full.extrema()[0].shape[0], len(full.keypoints()[0]), len(ereg),
           len(reg)))


if __name__ == "__main__":
    main()
