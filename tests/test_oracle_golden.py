"""The committed golden vectors (tests/golden/, produced by
tests/golden/make_golden.py from the reference's sample photograph) pin the
oracle against drift: it must keep reproducing them bit for bit."""
import numpy as np

import common


def test_crop_golden(oracle):
    g = np.load(common.GOLDEN + "/sunflower_crop.npz")
    gray = common.load_sunflower_gray()
    x0, y0, w, h = (int(v) for v in g["crop"])
    crop = np.ascontiguousarray(gray[y0:y0 + h, x0:x0 + w])
    assert common.sha(crop) == str(g["crop_sha256"])
    r = oracle.RefSift(crop, oracle.PyramidParams(0, 6, None, 1, 0.5, 1.6, 4))
    reg, so, desc = r.keypoints()
    ereg, exyso = r.extrema()
    assert np.array_equal(reg.view(np.uint8).reshape(-1, 48), g["regions"])
    assert np.array_equal(so, g["scale_octave"])
    assert np.array_equal(desc, g["descriptors"])
    assert np.array_equal(ereg.view(np.uint8).reshape(-1, 48), g["extrema"])
    assert np.array_equal(exyso, g["extrema_xyso_type"])
    names = [str(n) for n in g["plane_names"]]
    shas = [str(s) for s in g["plane_sha256"]]
    for name, want in zip(names, shas):
        kind, s, o = name.split("_")
        fn = {"G": r.gaussian, "D": r.dog, "grad": r.gradient}[kind]
        assert common.sha(fn(int(s), int(o))) == want, name


def test_full_frame_golden(oracle):
    g = np.load(common.GOLDEN + "/sunflower_full.npz")
    gray = common.load_sunflower_gray()
    assert gray.shape == (1200, 1600)
    assert common.sha(gray) == str(g["gray_sha256"])
    r = oracle.RefSift(gray, oracle.PyramidParams(0, 6, None, 1, 0.5, 1.6, 4),
                       parallel=True)
    reg, so, desc = r.keypoints()
    assert len(reg) == int(g["n_keypoints"]) == 6832
    assert len(r.extrema()[0]) == int(g["n_extrema"]) == 5592
    assert np.array_equal(reg.view(np.uint8).reshape(-1, 48), g["regions"])
    assert np.array_equal(so, g["scale_octave"])
    assert np.array_equal(desc[::8], g["desc_every8"])
    assert np.array_equal(desc.sum(axis=1), g["desc_row_sums"])
