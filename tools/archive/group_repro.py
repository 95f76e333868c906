#!/usr/bin/env python3
"""Three rounds of SiftGroup.detect over the loopback transport (the sequence of
tests/test_gpu_multi.py::test_loopback_group_gather_equals_single_context)."""
import os
import sys
os.environ.setdefault("SARA_HIP_COMM_TRANSPORT", "loopback")
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
if "--torch" in sys.argv:
    import torch  # noqa: F401
import sara_amd  # noqa: E402
import sara_amd.distributed as sd  # noqa: E402
from sara_amd.synth import synth_batch  # noqa: E402

frames = synth_batch(200, 160, 5, first_index=6)
p = sara_amd.ImagePyramidParams(0, 6, num_octaves_max=3)
if "--ref" in sys.argv:
    with sara_amd.SiftContext(200, 160, 5, p) as ctx:
        ctx.detect(frames)
        ctx.fetch()
g = sd.SiftGroup(200, 160, 3, p, n_dev=2)
for rnd in range(4):
    res = g.detect(frames if rnd % 2 == 0 else frames[::-1].copy())
    if rnd < 2:
        r = res.gather(root=rnd)
        print("round", rnd, r.total, flush=True)
    else:
        hc, hf, hd, hs = res.collect_host()
        print("round", rnd, hc, flush=True)
g.close()
print("done")
