import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import svtlib
    return svtlib.load_oracle()


@pytest.fixture(scope="session")
def product():
    """The HIP library.  No fallback: a missing build or device is a test failure."""
    import svtlib
    return svtlib.load_product()


@pytest.fixture(scope="session")
def gpu_ctx(product):
    import ctypes as C
    ctx = C.c_void_p()
    rc = product.svt_amd_context_create(0, 1920, 1088, 6, C.byref(ctx))
    assert rc == 0, product.svt_amd_last_error()
    yield ctx
    product.svt_amd_context_destroy(ctx)
