"""-m gpu: calls that stop at the Gaussian planes or the extremum sites replay
the fork-free level schedule (sift_detect.cpp): parity against the oracle.
Named to run FIRST among the GPU files: on runtimes before ROCm 7.2 a process
instantiates a bounded number of graphs, after which new
contexts run plain launches and would not reach this schedule."""
import numpy as np
import pytest

import common
import sara_amd
from sara_amd.synth import synth, synth_batch
from test_gpu_pipeline import (SHAPE_RTOL, compare_full, compare_lists, hip_params,
                               ref_params, run_lists)

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("w,h,first,noct,b", [
    (200, 150, 0, 4, 1), (257, 131, 0, 5, 1), (640, 480, 0, 4, 1),
    (97, 203, -1, 4, 1), (1920, 1080, 0, 4, 1), (333, 222, 0, 3, 3),
    (64, 48, 0, 2, 8), (150, 200, -1, 3, 2), (96, 80, 0, 3, 12), (120, 64, 0, 2, 16)])
def test_site_only_calls_take_the_level_schedule(oracle, w, h, first, noct, b):
    """A call of up to 8 frames that stops at the Gaussian planes or at the
    extremum sites replays the fork-free schedule (same-depth blurs of all
    octaves in one launch, all scans in one launch): planes, sites and their
    order are the oracle's, and a following full call on the same context (the
    forked layout) still gives the oracle's keypoints."""
    imgs = synth_batch(w, h, b) if b > 1 else synth(w, h, 21)[None]
    refs = [oracle.RefSift(im, ref_params(oracle, first, noct)) for im in imgs]
    S = refs[0].params.scale_count_per_octave
    with sara_amd.SiftContext(w, h, b, hip_params(first, noct)) as ctx:
        ctx.detect(imgs, last_stage=sara_amd.STAGE_PYRAMID)
        for f, ref in enumerate(refs):
            for o in range(ref.octave_count):
                for s in range(S):
                    assert np.array_equal(ctx.gaussian(s, o, f), ref.gaussian(s, o)), \
                        ("G", s, o, f)
        for _ in range(2):  # capture, then replay
            ctx.detect(imgs, last_stage=sara_amd.STAGE_EXTREMA)
            # a replayed graph has one timer (the schedule exists only there:
            # this file runs first, inside the graph budget of old runtimes)
            t = ctx.stage_times()
            assert t["total"] > 0 and t["pyramid"] == 0
            ec, ereg, exyso = ctx.extrema()
            want = [r.extrema()[1] for r in refs]
            assert [int(n) for n in ec[:b]] == [len(x) for x in want]
            assert np.array_equal(exyso, np.concatenate(want))
            for f, ref in enumerate(refs):
                for o in range(ref.octave_count):
                    for s in range(S - 1):
                        assert np.array_equal(ctx.dog(s, o, f), ref.dog(s, o)), \
                            ("D", s, o, f)
        ctx.detect(imgs)
        lists = run_lists(ctx)
        for f, ref in enumerate(refs):
            compare_full(ctx, ref, f, check_planes=False)
            compare_lists(lists, ref, f)


def _site_only_fuzz_cases(n, seed):
    """Random sizes and parameters inside the level schedule's conditions: six
    scales per octave, at least two octaves."""
    rng = np.random.default_rng(seed)
    cases = []
    for i in range(n):
        w = int(rng.integers(12, 120)) * 4 if i % 2 == 0 else int(rng.integers(40, 480))
        h = int(rng.integers(33, 400))
        first = int(rng.choice([0, 0, 0, -1])) if max(w, h) <= 200 else 0
        cam = float(rng.choice([0.5, 0.5, 1.0, 0.8]))
        noct = int(rng.integers(2, 7))
        thres = float(rng.choice([0.01, 0.01, 0.02, 0.005]))
        edge = float(rng.choice([10.0, 10.0, 6.0, 20.0]))
        iters = int(rng.choice([5, 5, 2, 3]))
        b = int(rng.choice([1, 1, 1, 2, 5]))
        cases.append((i, w, h, first, cam, noct, thres, edge, iters, b))
    return cases


@pytest.mark.parametrize("case", _site_only_fuzz_cases(16, 20260930),
                         ids=lambda c: "sites%d_%dx%dx%d" % (c[0], c[1], c[2], c[9]))
def test_fuzz_site_only_calls(oracle, case):
    i, w, h, first, cam, noct, thres, edge, iters, b = case
    imgs = np.stack([synth(w, h, 7000 + 16 * i + f) for f in range(b)])
    refs = [oracle.RefSift(im, ref_params(oracle, first, noct, cam),
                           extremum_thres=thres, edge_ratio_thres=edge,
                           extremum_refinement_iter=iters) for im in imgs]
    S = refs[0].params.scale_count_per_octave
    with sara_amd.SiftContext(w, h, b, hip_params(first, noct, cam),
                              extremum_thres=thres, edge_ratio_thres=edge,
                              extremum_refinement_iter=iters) as ctx:
        ctx.detect(imgs, last_stage=sara_amd.STAGE_EXTREMA)
        ec, ereg, exyso = ctx.extrema()
        want = [r.extrema() for r in refs]
        assert [int(n) for n in ec[:b]] == [len(x[1]) for x in want]
        assert np.array_equal(exyso, np.concatenate([x[1] for x in want]))
        e0 = 0
        for f, ref in enumerate(refs):
            common.assert_regions_equal(ereg[e0:e0 + len(want[f][0])], want[f][0],
                                        rtol_shape=SHAPE_RTOL)
            e0 += len(want[f][0])
            for o in range(ref.octave_count):
                for s in range(S):
                    assert np.array_equal(ctx.gaussian(s, o, f), ref.gaussian(s, o)), \
                        ("G", s, o, f)
