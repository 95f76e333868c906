import os
import sys

import pytest

# The library sends launches of fewer than 4 Mpx to the tiled blur (they cannot
# fill the chip with marching waves).  The parity tests use small images, so
# they lower that threshold to keep the production kernels - marching, fused
# pair, hand-scheduled - under test; test_gpu_full_size.py::test_default_small_
# launch_threshold re-runs one case in a subprocess with the shipped default.
os.environ.setdefault("SARA_HIP_MARCH_MIN_PIXELS", "0")
# Strip-group workgroups (8 / 4 adjacent strips under a barrier) are taken in
# production only by launches that fill the chip (>= 4096 / 2048 waves); the
# tests force them at every size for the same reason.  The shipped selection
# runs in test_gpu_full_size.py::test_shipped_kernel_selection_* (fresh
# processes without either variable).
os.environ.setdefault("SARA_HIP_STRIP_GROUP", "8")

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu)")


def _usable_cpus():
    """CPUs this process may use: affinity capped by the cgroup quota (the GPU
    boxes show 256 logical CPUs under a 16-CPU quota; an OpenMP team of 256 is
    about 100x slower than one of 16)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


@pytest.fixture(scope="session")
def oracle():
    import refbind
    refbind.build()
    refbind.lib().ref_omp_set_threads(_usable_cpus())
    return refbind


@pytest.fixture(params=["forced", "shipped"])
def kernel_selection(request):
    """Round 6: the same test under the kernel selection the environment above
    forces (marching kernels and 8-strip groups at every launch size) AND under
    the one the library ships (thresholds as bench.py and every user run them),
    in one process: SARA_HIP_OPT_KERNEL_SELECTION is a context option, applied
    to every context the test creates - the cached ones of the free functions
    included (sara_amd.DEFAULT_OPTIONS)."""
    import sara_amd
    from sara_amd import capi
    # The shipped pass runs plain launches: which kernels run does not depend on
    # how they are submitted, graph replay of the shipped selection has its own
    # fresh-process tests (test_gpu_full_size.py), and the ROCm 7.0 runtime that
    # `import torch` loads does not survive many more graph captures per process
    # than round 5's suite already made (graph_launcher.cpp: kOldRuntimeGraphBudget).
    options = ({capi.OPT_KERNEL_SELECTION: capi.SELECT_FORCED_MARCH}
               if request.param == "forced" else
               {capi.OPT_KERNEL_SELECTION: capi.SELECT_SHIPPED,
                capi.OPT_GRAPH_REPLAY: 0})
    with sara_amd.default_options(options):
        yield request.param
