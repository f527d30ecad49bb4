import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with gpurun / at round end)")


@pytest.fixture(scope="session")
def sa():
    """The product: vaex_amd.superagg (pybind11 shim over the C-ABI of libvaexhip.so)."""
    import vaex_amd
    return vaex_amd.superagg


@pytest.fixture(scope="session")
def ref():
    """The reference's own superagg C++, compiled in place into oracle/_ref (skips when absent)."""
    from oracle import oracle
    m = oracle.ref_module("superagg")
    if m is None:
        pytest.skip("oracle/_ref/superagg not built")
    return m


@pytest.fixture(scope="session")
def gpu_ready(sa):
    if sa.device_count() == 0:
        pytest.fail("no HIP device visible: -m gpu tests need the GPU box")
    return True


import contextlib


@contextlib.contextmanager
def knob(sa, key, value):
    """vxh_config_set(key, value) for the duration of the block, then the value it had before (the library's defaults are the
    reference's behaviour: first_mask_block = 1024, nunique_row_counts = 1)"""
    before = sa.config_get(key)
    sa.config_set(key, value)
    try:
        yield
    finally:
        sa.config_set(key, before)
