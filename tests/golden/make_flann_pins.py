"""Generates tests/golden/flann_pins.npz (run in the build container, where
/root/reference exists; no test reads /root/reference).

What it records: the answers of the reference's OWN vendored FLANN
(/root/reference/cpp/third-party/flann, header-only, compiled by
`make -C oracle _ref` into oracle/_ref/libflann_ref.so and called as
FeatureMatching/AnnMatcher.cpp calls it) on

  pair   the oracle's SIFT descriptors of two overlapping synthetic 1080p
         views (crops of one 1944 x 1088 scene, shifted by (24, 8) pixels -
         the matcher workload of bench.py), 4 octaves
  crop   the 791 descriptors / OERegions of the committed sunflower crop
         (tests/golden/sunflower_crop.npz), for the self-matching constructor

with both index kinds:
  linear   flann::LinearIndexParams - FLANN's exact search: knnSearch(3),
           radiusSearch(d_best * 1.2^2) and the final compute_matches() lists;
           the exhaustive oracle and the GPU matcher must equal these BIT FOR BIT
  kdtree   flann::KDTreeIndexParams(8), 32 checks, seed 0 - the reference's
           real configuration (AnnMatcher.cpp:227): approximate; recorded so
           that the distance between "exact" and "what Sara returns" is a number

The descriptors themselves are not stored (the tests regenerate them with the
oracle and check their SHA-256 against `pair_sha256`)."""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import flannbind as fb  # noqa: E402
import refbind as rb  # noqa: E402

RATIOS = (0.6, 0.8, 1.0, 1.2)


def sha(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def pair_descriptors():
    """-> (desc1, desc2) of the two views; shared with the tests."""
    from sara_amd.synth import synth
    W, H = 1920, 1080
    scene = synth(W + 24, H + 8, 1234)
    views = (np.ascontiguousarray(scene[:H, :W]),
             np.ascontiguousarray(scene[8:H + 8, 24:W + 24]))
    P = rb.PyramidParams(0, 6, None, 1, 0.5, 1.6, 4)
    return tuple(rb.RefSift(v, P, parallel=True).keypoints()[2] for v in views)


def ragged(rows):
    off = np.concatenate([[0], np.cumsum([len(i) for i, _ in rows])]).astype(np.int64)
    idx = np.concatenate([i for i, _ in rows]).astype(np.int32)
    dist = np.concatenate([d for _, d in rows]).astype(np.float32)
    return off, idx, dist


def main():
    d1, d2 = pair_descriptors()
    out = {"pair_sha256": np.array(sha(d1, d2)),
           "pair_counts": np.array([len(d1), len(d2)]),
           "ratios": np.array(RATIOS, np.float32)}
    for kind, name in ((fb.LINEAR, "linear"), (fb.KDTREE8, "kdtree")):
        for tag, (q, t) in (("12", (d1, d2)), ("21", (d2, d1))):
            idx, dist = fb.knn(t, q, 3, kind)
            out["%s_knn3_idx_%s" % (name, tag)] = idx
            out["%s_knn3_dist_%s" % (name, tag)] = dist
        for r in RATIOS:
            out["%s_matches_%.1f" % (name, r)] = fb.compute_matches(d1, d2, r, kind)
    # AnnMatcher.cpp:133-138: radius = d_best * ratio^2 (float product)
    best = out["linear_knn3_dist_12"][:, 0]
    radii = (best * np.float32(np.float32(1.2) * np.float32(1.2))).astype(np.float32)
    off, idx, dist = ragged(fb.radius(d2, d1, radii, fb.LINEAR))
    out.update(linear_radius_off_12=off, linear_radius_idx_12=idx,
               linear_radius_dist_12=dist, linear_radius_r_12=radii)

    crop = np.load(os.path.join(HERE, "sunflower_crop.npz"))
    cd = crop["descriptors"]
    creg = np.ascontiguousarray(crop["regions"]).view(rb.OEREGION_DTYPE).reshape(-1)
    out["crop_sha256"] = np.array(sha(cd, crop["regions"]))
    for kind, name in ((fb.LINEAR, "linear"), (fb.KDTREE8, "kdtree")):
        out["%s_self_matches_crop" % name] = fb.compute_self_matches(
            cd, creg, 1.2, 0.5, 10.0, kind)
        idx, dist = fb.knn(cd, cd, 3, kind)
        out["%s_knn3_idx_crop" % name] = idx
        out["%s_knn3_dist_crop" % name] = dist
    np.savez_compressed(os.path.join(HERE, "flann_pins.npz"), **out)
    for k, v in out.items():
        print(k, v.shape, v.dtype)


if __name__ == "__main__":
    main()
