"""Generates the real-image parity pack (run in the build container, where
/root/reference exists; no test reads /root/reference).

Inputs are DATA: the photographs the reference's own examples and tests use
(/root/reference/data), decoded here to RGB8 and stored losslessly as PNG under
tests/golden/real/, so that the GPU box depends on no TIFF / JPEG decoder:

  All.tif, GuardOnBlonde.tif   THE matching pair of
                               examples/Sara/FeatureMatching/image_sift_matching.cpp:25-26
  sift_edge.jpg                a 319 x 67 strip of straight edges (on_edge, plateaus)
  ksmall.jpg, dog.jpg          JPEG photographs (8 x 8 block structure, saturation)
  stinkbug.png                 a PNG photograph with a flat saturated background
  image-pinhole.png, image-omni.png
                               two real 1920 x 1080 camera frames (pinhole and
                               fisheye): the benchmark's frame size on real input

Expected outputs come from the CPU oracle (oracle/sift_ref.hpp, pinned by the
reference's unit tests) with two parameter sets:

  default   ImagePyramidParams() as the C++ example calls it: first octave -1
            (the frame is enlarged 2 x), every octave the size allows
  bench     first octave 0, 4 octaves (the benchmark's / FeatureParams' set)

and, for the pair, from the reference's OWN vendored FLANN (exact linear
index, oracle/_ref/libflann_ref.so called as FeatureMatching/AnnMatcher.cpp
calls it) on the oracle's `default` descriptors: compute_matches() at
sift_ratio_thres 0.6, 1.0 (the example's value, :57) and 1.2 (AnnMatcher's
default).

  real/<name>.png         RGB8
  real_images.npz         per image and parameter set: gray_sha256, regions
                          (N x 48 bytes), scale_octave, extrema_xyso_type,
                          descriptors of every 8th keypoint, row sums of all;
                          pair_sha256 (the two descriptor matrices) and
                          pair_matches_<ratio>
"""
import hashlib
import os
import sys

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import flannbind as fb  # noqa: E402
import refbind as rb  # noqa: E402

DATA = "/root/reference/data"
IMAGES = ("All.tif", "GuardOnBlonde.tif", "sift_edge.jpg", "ksmall.jpg",
          "dog.jpg", "stinkbug.png", "image-pinhole.png", "image-omni.png")
PAIR = ("All", "GuardOnBlonde")
RATIOS = (0.6, 1.0, 1.2)


def param_sets():
    return {"default": rb.PyramidParams(),
            "bench": rb.PyramidParams(0, 6, None, 1, 0.5, 1.6, 4)}


def sha(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def main():
    os.makedirs(os.path.join(HERE, "real"), exist_ok=True)
    out = {"names": np.array([os.path.splitext(f)[0] for f in IMAGES]),
           "ratios": np.array(RATIOS, np.float32)}
    pair_desc = {}
    for f in IMAGES:
        name = os.path.splitext(f)[0]
        rgb = np.array(Image.open(os.path.join(DATA, f)).convert("RGB"))
        Image.fromarray(rgb).save(os.path.join(HERE, "real", name + ".png"),
                                  optimize=True)
        gray = rb.rgb8_to_gray32f(rgb)
        out[name + "_gray_sha256"] = np.array(sha(gray))
        for tag, params in param_sets().items():
            r = rb.RefSift(gray, params, parallel=True)
            reg, so, desc = r.keypoints()
            ereg, exyso = r.extrema()
            k = "%s_%s_" % (name, tag)
            out[k + "regions"] = reg.view(np.uint8).reshape(-1, 48)
            out[k + "scale_octave"] = so
            out[k + "extrema_xyso_type"] = exyso
            out[k + "extrema_regions"] = ereg.view(np.uint8).reshape(-1, 48)
            out[k + "desc_every8"] = desc[::8]
            out[k + "desc_row_sums"] = desc.sum(axis=1)
            out[k + "octaves"] = np.array(r.octave_count)
            print("%-14s %-8s %4d x %-4d octaves %d extrema %5d keypoints %5d"
                  % (name, tag, rgb.shape[1], rgb.shape[0], r.octave_count,
                     len(ereg), len(reg)))
            if name in PAIR and tag == "default":
                pair_desc[name] = desc
    d1, d2 = pair_desc[PAIR[0]], pair_desc[PAIR[1]]
    out["pair_sha256"] = np.array(sha(d1, d2))
    for ratio in RATIOS:
        m = fb.compute_matches(d1, d2, ratio, fb.LINEAR)
        out["pair_matches_%.1f" % ratio] = m
        print("pair, ratio %.1f: %d matches (FLANN linear)" % (ratio, len(m)))
    np.savez_compressed(os.path.join(HERE, "real_images.npz"), **out)


if __name__ == "__main__":
    main()
