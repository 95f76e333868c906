"""Distance oracle -> reference, measured: what does the last-ulp arithmetic of
make_gaussian_kernel (ImageProcessing/LinearFiltering.hpp:171-203: Eigen's
packet exp() and vectorised sum(), which cannot be compiled on this image) do
to the keypoints?

    python tests/golden/make_sensitivity.py            # writes sensitivity.json
    python tests/golden/make_sensitivity.py --quick    # two workloads, stdout

For every workload - the eight real images of the parity pack x two parameter
sets, and 64 synthetic 1080p frames with the benchmark's parameters - the CPU
oracle (oracle/sift_ref.hpp) runs once with its default taps (expf + serial
sum, the arithmetic the HIP product's host code shares) and once per tap
variant (sift_ref.hpp kTaps*: the Eigen 3.4 / 3.3 SSE2 models of what the
reference's Release build evaluates, and one-ulp perturbations of every tap:
three random sign patterns, all up, all down, alternating, narrow, wide).

Reported per variant (summed over the workloads, and per workload):
  * extremum level (before orientations): sites are identified by their
    integer DoG coordinates (x, y, s, o); lost / gained / type flips; drift of
    the survivors' refined coordinates (image px), sigma (relative), value;
  * keypoint level: a keypoint of the base run is "kept" when the variant has
    one with the same (s, o), within 0.5 px and within half an orientation bin
    (pi / 36); lost / gained otherwise; for the kept ones the drift of the
    coordinates, sigma, orientation, and the descriptors' max-abs difference
    (range 0..255) and worst L2 distance.

TEST INFRASTRUCTURE (imports the oracle).  tests/test_oracle_sensitivity.py
asserts the envelope stated in DESIGN.md section 5 on the committed file and
re-derives two of its rows."""
import argparse
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)
ROOT = os.path.dirname(TESTS)
for p in (ROOT, TESTS):
    if p not in sys.path:
        sys.path.insert(0, p)

import refbind as rb  # noqa: E402

# (name, kind, seed) - kinds are sift_ref.hpp's kTaps* values
VARIANTS = (
    ("eigen34_sse", 1, 0),
    ("eigen33_sse", 2, 0),
    ("ulp_random_1", 3, 1),
    ("ulp_random_2", 3, 2),
    ("ulp_random_3", 3, 3),
    ("ulp_plus", 4, 0),
    ("ulp_minus", 5, 0),
    ("ulp_alternate", 6, 0),
    ("ulp_narrow", 7, 0),
    ("ulp_wide", 8, 0),
)
# the Gaussian blurs of the default schedule
SCHEDULE_SIGMAS = (1.51986849, 1.22627354, 1.54500782, 1.94658804, 2.45254707,
                   3.09001565, 1.24899971)
HALF_BIN = float(np.pi / 36)
MATCH_PX = 0.5


def set_variant(kind, seed):
    lib = rb.lib()
    lib.ref_set_tap_variant.argtypes = [rb.C.c_int, rb.C.c_uint]
    lib.ref_set_tap_variant.restype = rb.C.c_int
    lib.ref_set_tap_variant(kind, seed)


def run(image, params):
    r = rb.RefSift(image, params, parallel=True)
    ext, xyso = r.extrema()
    reg, so, desc = r.keypoints()
    return dict(ext=ext, xyso=xyso, reg=reg, so=so, desc=desc,
                factors=[r.octave_info(o)[2] for o in range(r.octave_count)])


def sigma_of(regions):
    # shape matrix = I / sigma^2 (RefineExtremum.cpp:497-515, SIFT.cpp:92-98)
    return 1.0 / np.sqrt(regions["shape_matrix"][:, 0].astype(np.float64))


def compare_extrema(base, var):
    """Sites by integer DoG coordinates.  The extrema are in octave
    coordinates; drifts are reported in image pixels (x octave factor)."""
    def keys(x):
        a = x["xyso"].astype(np.int64)
        return (a[:, 3] << 48) | (a[:, 2] << 40) | (a[:, 1] << 20) | a[:, 0]
    kb, kv = keys(base), keys(var)
    common, ib, iv = np.intersect1d(kb, kv, return_indices=True)
    out = dict(base=int(len(kb)), variant=int(len(kv)),
               lost=int(len(kb) - len(common)),
               gained=int(len(kv) - len(common)))
    if len(common):
        eb, ev = base["ext"][ib], var["ext"][iv]
        f = np.asarray(base["factors"], np.float64)[base["xyso"][ib, 3]]
        d = np.abs(eb["coords"].astype(np.float64) - ev["coords"]).max(axis=1) * f
        sb, sv = sigma_of(eb), sigma_of(ev)
        out.update(
            type_flips=int(np.sum(eb["extremum_type"] != ev["extremum_type"])),
            moved=int(np.sum(d > 0)),
            max_xy_px=float(d.max()),
            max_sigma_rel=float(np.max(np.abs(sv / sb - 1.0))),
            max_value_abs=float(np.max(np.abs(
                eb["extremum_value"].astype(np.float64) - ev["extremum_value"]))))
    return out


def compare_keypoints(base, var):
    from scipy.spatial import cKDTree
    rb_, rv = base["reg"], var["reg"]
    out = dict(base=int(len(rb_)), variant=int(len(rv)))
    kept_b, kept_v = [], []
    so_b = base["so"][:, 0].astype(np.int64) * 64 + base["so"][:, 1]
    so_v = var["so"][:, 0].astype(np.int64) * 64 + var["so"][:, 1]
    for key in np.unique(so_b):
        jb = np.flatnonzero(so_b == key)
        jv = np.flatnonzero(so_v == key)
        if not len(jv):
            continue
        tree = cKDTree(rv["coords"][jv].astype(np.float64))
        used = set()
        dist, idx = tree.query(rb_["coords"][jb].astype(np.float64), k=4,
                               distance_upper_bound=MATCH_PX * np.sqrt(2) + 1e-9)
        dist = np.atleast_2d(dist)
        idx = np.atleast_2d(idx)
        for row in range(len(jb)):
            i = jb[row]
            best, best_d = -1, None
            for d, j in zip(dist[row], idx[row]):
                if not np.isfinite(d) or j >= len(jv):
                    continue
                jj = jv[j]
                if jj in used:
                    continue
                dxy = np.abs(rb_["coords"][i].astype(np.float64) -
                             rv["coords"][jj]).max()
                dth = abs(float(rb_["orientation"][i]) -
                          float(rv["orientation"][jj]))
                dth = min(dth, 2 * np.pi - dth)
                if dxy <= MATCH_PX and dth <= HALF_BIN:
                    score = (dxy, dth)
                    if best_d is None or score < best_d:
                        best, best_d = jj, score
            if best >= 0:
                used.add(best)
                kept_b.append(i)
                kept_v.append(best)
    kb = np.asarray(kept_b, np.int64)
    kv = np.asarray(kept_v, np.int64)
    out.update(kept=int(len(kb)), lost=int(len(rb_) - len(kb)),
               gained=int(len(rv) - len(kv)))
    if len(kb):
        a, b = rb_[kb], rv[kv]
        dxy = np.abs(a["coords"].astype(np.float64) - b["coords"]).max(axis=1)
        dth = np.abs(a["orientation"].astype(np.float64) - b["orientation"])
        dth = np.minimum(dth, 2 * np.pi - dth)
        dd = np.abs(base["desc"][kb].astype(np.float64) - var["desc"][kv])
        l2 = np.sqrt((dd ** 2).sum(axis=1))
        out.update(
            identical=int(np.sum((dxy == 0) & (dth == 0) & (dd.max(axis=1) == 0) &
                                 (a["shape_matrix"] == b["shape_matrix"]).all(axis=1))),
            max_xy_px=float(dxy.max()),
            max_sigma_rel=float(np.max(np.abs(sigma_of(b) / sigma_of(a) - 1.0))),
            max_theta_rad=float(dth.max()),
            desc_max_abs=float(dd.max()),
            desc_p999_abs=float(np.quantile(dd.max(axis=1), 0.999)),
            desc_max_l2=float(l2.max()),
            desc_p999_l2=float(np.quantile(l2, 0.999)))
    return out


def merge(rows):
    """Sum the counts, max the drifts."""
    tot = {}
    for r in rows:
        for level in ("extrema", "keypoints"):
            t = tot.setdefault(level, {})
            for k, v in r[level].items():
                if k.startswith("max_") or k.startswith("desc_"):
                    t[k] = max(t.get(k, 0.0), v)
                else:
                    t[k] = t.get(k, 0) + v
    for level in ("extrema", "keypoints"):
        t = tot[level]
        t["lost_frac"] = t["lost"] / max(1, t["base"])
        t["gained_frac"] = t["gained"] / max(1, t["base"])
    return tot


def tap_table():
    """Taps of the default schedule under each variant, as ulp offsets from
    the default arithmetic."""
    set_variant(0, 0)
    base = {s: rb.make_gaussian_kernel(s) for s in SCHEDULE_SIGMAS}
    table = {}
    for name, kind, seed in VARIANTS:
        set_variant(kind, seed)
        rows = {}
        for s in SCHEDULE_SIGMAS:
            k = rb.make_gaussian_kernel(s)
            d = (k.view(np.int32).astype(np.int64) - base[s].view(np.int32))
            rows["%.8f" % s] = dict(
                ulp=[int(v) for v in d],
                symmetric=bool(np.array_equal(k, k[::-1])),
                sum_minus_1=float(k.astype(np.float64).sum() - 1.0))
        table[name] = rows
    set_variant(0, 0)
    return table


def workloads(quick):
    import real_images as ri
    from sara_amd.synth import synth
    items = []
    names = ("ksmall",) if quick else ri.NAMES
    for name in names:
        for tag in (("bench",) if quick else ri.TAGS):
            items.append(("real/%s/%s" % (name, tag),
                          (lambda n=name: ri.gray(rb, n)),
                          ri.ref_params(rb, tag)))
    bench = rb.PyramidParams(0, 6, None, 1, 0.5, 1.6, 4)
    for i in range(1 if quick else 64):
        items.append(("synth1080p/%d" % i,
                      (lambda k=i: synth(1920, 1080, 1234 + k)), bench))
    return items


def measure(quick=False, variants=VARIANTS, log=None):
    per_variant = {name: [] for name, _, _ in variants}
    rows = []
    for label, load, params in workloads(quick):
        t0 = time.time()
        img = load()
        set_variant(0, 0)
        base = run(img, params)
        for name, kind, seed in variants:
            set_variant(kind, seed)
            var = run(img, params)
            row = dict(workload=label, variant=name,
                       extrema=compare_extrema(base, var),
                       keypoints=compare_keypoints(base, var))
            per_variant[name].append(row)
            rows.append(row)
        set_variant(0, 0)
        if log:
            log("%-28s %6d keypoints  %.1f s" % (label, len(base["reg"]),
                                                 time.time() - t0))
    groups = {}
    for name, items in per_variant.items():
        groups[name] = dict(
            all=merge(items),
            real=merge([r for r in items if r["workload"].startswith("real/")] or items),
            synth1080p=merge([r for r in items
                              if r["workload"].startswith("synth")] or items))
    return groups, rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--out", default=os.path.join(HERE, "sensitivity.json"))
    args = ap.parse_args()
    groups, rows = measure(args.quick, log=lambda s: print(s, flush=True))
    doc = dict(
        what="keypoints of the CPU oracle under tap-arithmetic variants of "
             "make_gaussian_kernel vs its default (expf + serial sum); see "
             "tests/golden/make_sensitivity.py",
        match_rule=dict(px=MATCH_PX, theta_rad=HALF_BIN,
                        extrema="same integer DoG site (x, y, s, o)"),
        taps_ulp_vs_default=tap_table(),
        per_variant=groups,
        rows=rows)
    if args.quick:
        print(json.dumps(groups, indent=1))
        return
    # one line per (workload, variant) row, the rest indented
    rows = doc.pop("rows")
    head = json.dumps(doc, indent=1, sort_keys=True)
    body = ",\n".join("  " + json.dumps(r, sort_keys=True, separators=(",", ":"))
                      for r in rows)
    with open(args.out, "w") as f:
        f.write(head[:-2] + ',\n "rows": [\n' + body + "\n ]\n}\n")
    for name, g in groups.items():
        k, e = g["all"]["keypoints"], g["all"]["extrema"]
        print("%-14s extrema lost %d gained %d of %d | keypoints lost %d gained "
              "%d of %d | kept: xy %.3g px theta %.3g rad desc %.3g" % (
                  name, e["lost"], e["gained"], e["base"], k["lost"], k["gained"],
                  k["base"], k.get("max_xy_px", 0), k.get("max_theta_rad", 0),
                  k.get("desc_max_abs", 0)))


if __name__ == "__main__":
    main()
