"""The distance ORACLE -> REFERENCE, as far as it can be measured without the
reference (Eigen is not on the image): tests/golden/sensitivity.json holds what
the keypoints of the real-image pack (8 images x 2 parameter sets) and of 64
synthetic 1080p frames do when make_gaussian_kernel's two non-IEEE operations -
Eigen's array exp() and sum(), LinearFiltering.hpp:196-200 - are evaluated the
way the reference's Release build evaluates them (Eigen 3.4 / 3.3 SSE2 packet
models) or when every tap moves by one ulp (tests/golden/make_sensitivity.py).

This file asserts the ENVELOPE DESIGN.md section 5 states as the tolerance to
the reference, re-derives rows of the committed file from the code, and shows
that the GPU's bars against the oracle sit far inside the envelope."""
import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))

import make_sensitivity as ms  # noqa: E402

# DESIGN.md section 5, "tolerance to the reference"
MAX_CHANGED_FRACTION = 3e-4      # keypoints lost (or gained) / keypoints
KEPT_DESC_P999 = 6.0             # of 255: 99.9 % of the kept keypoints
KEPT_SIGMA_REL = 0.03
KEPT_THETA = ms.HALF_BIN         # by the match rule
KEPT_XY = ms.MATCH_PX            # by the match rule


@pytest.fixture(scope="module")
def doc():
    return json.load(open(os.path.join(HERE, "golden", "sensitivity.json")))


def test_the_file_covers_the_pack_and_every_variant(doc):
    import real_images as ri
    names = {name for name, _, _ in ms.VARIANTS}
    assert set(doc["per_variant"]) == names
    workloads = {r["workload"] for r in doc["rows"]}
    want = {"real/%s/%s" % (n, t) for n in ri.NAMES for t in ri.TAGS}
    want |= {"synth1080p/%d" % i for i in range(64)}
    assert workloads == want
    assert len(doc["rows"]) == len(want) * len(names)
    # what the variants are: ulp offsets of the default schedule's taps
    taps = doc["taps_ulp_vs_default"]
    assert all(set(u) <= {-1, 0, 1} for v in taps if v.startswith("ulp_")
               for row in taps[v].values() for u in [row["ulp"]])
    assert max(abs(u) for row in taps["eigen34_sse"].values()
               for u in row["ulp"]) == 2
    # the packet / scalar split of the dense assignment loop makes some of
    # Eigen's kernels asymmetric (sigma 1.545: 13 taps = 12 packet + 1 scalar)
    assert not taps["eigen34_sse"]["1.54500782"]["symmetric"]
    assert taps["eigen34_sse"]["1.22627354"]["symmetric"]


@pytest.mark.parametrize("group", ["all", "real", "synth1080p"])
def test_envelope(doc, group):
    """Whatever the reference's exp() / sum() do within an ulp per tap:
    at most 3 keypoints in 10 000 come or go, no extremum changes its type,
    and the kept ones stay within the stated drifts."""
    seen = 0
    for name, g in doc["per_variant"].items():
        k, e = g[group]["keypoints"], g[group]["extrema"]
        assert k["base"] > 18000
        assert k["lost"] <= MAX_CHANGED_FRACTION * k["base"], (name, k["lost"])
        assert k["gained"] <= MAX_CHANGED_FRACTION * k["base"], (name, k["gained"])
        assert e["lost"] <= MAX_CHANGED_FRACTION * e["base"]
        assert e["gained"] <= MAX_CHANGED_FRACTION * e["base"]
        assert e["type_flips"] == 0
        assert k["kept"] + k["lost"] == k["base"]
        assert k["max_xy_px"] <= KEPT_XY and k["max_theta_rad"] <= KEPT_THETA
        assert k["max_sigma_rel"] <= KEPT_SIGMA_REL
        assert k["desc_p999_abs"] <= KEPT_DESC_P999
        seen += k["lost"] + k["gained"]
    # ... and the arithmetic is NOT invisible: the hard decisions of the
    # detector (>= / < at every stage) do flip for a few sites
    assert seen > 0


def test_the_change_is_real_but_rare_per_frame(doc):
    """Per 1080p frame (about 4 380 keypoints): the Eigen 3.4 model changes at
    most a handful, most frames none."""
    rows = [r for r in doc["rows"] if r["variant"] == "eigen34_sse"
            and r["workload"].startswith("synth1080p/")]
    changed = [r["keypoints"]["lost"] + r["keypoints"]["gained"] for r in rows]
    assert len(rows) == 64 and max(changed) <= 8
    assert sum(c == 0 for c in changed) >= 16


def test_rows_re_derived_from_the_code(doc):
    """Two workloads x two variants computed again here equal the committed
    rows: the file belongs to this oracle and this script."""
    import real_images as ri
    import refbind as rb
    from sara_amd.synth import synth
    cases = [("real/ksmall/bench", lambda: ri.gray(rb, "ksmall"),
              ri.ref_params(rb, "bench")),
             ("synth1080p/3", lambda: synth(1920, 1080, 1234 + 3),
              rb.PyramidParams(0, 6, None, 1, 0.5, 1.6, 4))]
    variants = [v for v in ms.VARIANTS if v[0] in ("eigen34_sse", "ulp_minus")]
    committed = {(r["workload"], r["variant"]): r for r in doc["rows"]}
    try:
        for label, load, params in cases:
            img = load()
            ms.set_variant(0, 0)
            base = ms.run(img, params)
            for name, kind, seed in variants:
                ms.set_variant(kind, seed)
                var = ms.run(img, params)
                want = committed[(label, name)]
                got = dict(extrema=ms.compare_extrema(base, var),
                           keypoints=ms.compare_keypoints(base, var))
                for level in ("extrema", "keypoints"):
                    for key, value in want[level].items():
                        if isinstance(value, float):
                            assert got[level][key] == pytest.approx(value, rel=1e-9)
                        else:
                            assert got[level][key] == value, (label, name, key)
    finally:
        ms.set_variant(0, 0)


def test_gpu_bars_sit_inside_the_envelope(doc):
    """The HIP path equals the oracle (same taps) exactly in sites, order and
    coordinates, within 1e-6 rad and 2e-3 of 255 (tests/test_gpu_pipeline.py);
    the oracle itself is only known to equal the reference up to the envelope
    above, whose typical drifts are three orders of magnitude larger: the GPU
    adds nothing measurable to the distance to the reference."""
    gpu_theta, gpu_desc = 1e-6, 2e-3     # THETA_ATOL, DESC_ATOL of the GPU tests
    e34 = doc["per_variant"]["eigen34_sse"]["all"]["keypoints"]
    assert gpu_desc * 1000 <= e34["desc_p999_abs"]
    assert gpu_theta * 1000 <= e34["max_theta_rad"]
    # fewer than 1 keypoint in 100 keeps all its bits (those of the octaves
    # whose planes the changed taps happen to leave alone)
    assert e34["identical"] < 0.01 * e34["base"]
    assert e34["max_xy_px"] > 1e-3        # vs exact coordinates GPU <-> oracle
