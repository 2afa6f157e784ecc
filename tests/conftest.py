import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with `-m gpu` through gpurun)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU restatement (test infrastructure).  Built on demand with gcc."""
    from oracle import oracle as orc
    orc.build()
    return orc


@pytest.fixture(autouse=True)
def _process_wide_switches_back_to_default(request):
    """some implementation switches of libwlhip are process-wide (include/wlhip.h: wl_reset_process_options): a test that moves a size gate
    must not decide which kernels the next test runs"""
    yield
    if request.node.get_closest_marker("gpu") is not None:
        import waterlily_jl_amd as w
        w.lib().wl_reset_process_options()
