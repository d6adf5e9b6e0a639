import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# The suite runs on libtsdf_hip_test.so: the product's sources and kernels + the hooks of include/tsdf_hip_test.h (knobs that
# force a kernel instance, device-side dividers on chosen operands, host-side cull predicates).  The product library itself
# (libtsdf_hip.so, no hooks) is what the C++ drop-in links -- tests/test_dropin_gpu.py, tests/test_programs_gpu.py -- and what
# tests/test_product_lib_gpu.py, __graft_entry__.smoke() and bench.py load.
from cpu_tsdf_amd import capi  # noqa: E402

capi.use_test_library()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def _gpu_available():
    try:
        from cpu_tsdf_amd import capi
        return capi.load().tsdf_hip_device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu():
    """GPU tests must run the HIP path; on a GPU box a missing library is a failure, not a skip."""
    from cpu_tsdf_amd import capi
    lib = capi.load()
    if lib.tsdf_hip_device_count() <= 0:
        pytest.skip("no HIP device visible")
    return lib
