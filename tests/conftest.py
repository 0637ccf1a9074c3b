import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


# single-node RCCL in the tests: bootstrap over the loopback interface (the container's hostname may not resolve and
# interface probing has been seen to stall for minutes on some boxes), no InfiniBand probing.  The caller's settings win.
os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
os.environ.setdefault("NCCL_IB_DISABLE", "1")
# Block pivoting at k in (32, 64] on a matrix of at most 2^24 entries takes the accurate product form by default (solver.cpp).
# The suite's small cases stand in for the fp16 two-term path that C4 takes and that cannot be compared with the oracle at
# full size, so the suite keeps that form; test_gpu_variants.py::test_small_bpp_takes_the_accurate_form checks the default.
os.environ.setdefault("SMK_BPP_SMALL_ACCURATE", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a fresh checkout has no built artefacts: build them once (hipcc cross-compiles without a GPU)
    need = [os.path.join(ROOT, "smallk_amd", "lib", "libsmallk_amd.so"),
            os.path.join(ROOT, "smallk_amd", "bin", "nmf"),
            os.path.join(ROOT, "smallk_amd", "bin", "hierclust"),
            os.path.join(ROOT, "smallk_amd", "bin", "flatclust"),
            os.path.join(ROOT, "oracle", "_build", "liboracle.so")]
    if not all(os.path.exists(p) for p in need):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "nmf_golden.npz"))


@pytest.fixture(scope="session")
def gpu():
    """Initialise the HIP library once per session; fail loudly if it cannot run."""
    import smallk_amd
    smallk_amd.initialize(0)
    yield smallk_amd
