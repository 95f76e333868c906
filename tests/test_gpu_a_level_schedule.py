"""-m gpu: calls that stop at the Gaussian planes or the extremum sites replay
the fork-free level schedule (sift_detect.cpp): parity against the oracle.
Named to run FIRST among the GPU files: on runtimes before ROCm 7.2 a process
instantiates a bounded number of graphs, after which new
contexts run plain launches and would not reach this schedule."""
import numpy as np
import pytest

import sara_amd
from sara_amd.synth import synth, synth_batch
from test_gpu_pipeline import (compare_full, compare_lists, hip_params, ref_params,
                               run_lists)

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
