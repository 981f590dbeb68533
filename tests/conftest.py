import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "timeout: per-test time limit (pytest-timeout)")
    config.addinivalue_line("markers", "fast_blend: runs the rasterizer in its default (fast blend) mode; every other test runs the exact blend")


def pytest_collection_modifyitems(config, items):
    """No test of this suite needs minutes: a kernel that never returns (an out-of-bounds write did that once: 25 GPU-minutes) must fail the
    run, not hang it.  pytest-timeout is part of the image; without it the marker is inert."""
    if not config.pluginmanager.hasplugin("timeout"):
        return
    for item in items:
        if item.get_closest_marker("timeout") is None:
            item.add_marker(pytest.mark.timeout(600 if item.get_closest_marker("gpu") is None else 300, method="thread"))   # thread: a wait inside a native call never returns to the interpreter


@pytest.fixture(autouse=True)
def _exact_blend_unless_marked(request):
    """The parity bar of this suite is bit-exactness against the oracle, which is the EXACT blend (GsrSettings.fast_blend = 0).  The
    product default is the fast blend (same integers, image within a stated tolerance): tests marked `fast_blend` run that mode and
    state their tolerance; everything else selects the exact kernels for its duration."""
    if request.node.get_closest_marker("fast_blend") is not None:
        yield
        return
    from gaussianavatars_amd import rasterizer

    prev = rasterizer.set_fast_blend(False)
    try:
        yield
    finally:
        rasterizer.set_fast_blend(prev)


@pytest.fixture(scope="session")
def oracle():
    """The CPU restatement (test infrastructure).  Built on demand with its own Makefile."""
    from oracle import gsr_oracle

    gsr_oracle.build()
    return gsr_oracle
