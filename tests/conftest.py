import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Make sure the product library and the oracle are built (hipcc cross-compiles without a GPU)."""
    import subprocess
    need = [os.path.join(ROOT, "offt_amd", "liboffthip.so"), os.path.join(ROOT, "oracle", "liboracle.so"),
            os.path.join(ROOT, "tests", "libcpubackend.so"), os.path.join(ROOT, "tests", "liboffthip_test.so")]
    if not all(os.path.exists(n) for n in need):
        subprocess.check_call(["make", "-C", ROOT, "all", "tests/libcpubackend.so", "tests/liboffthip_test.so"])
    return True


@pytest.fixture(autouse=True)
def _release_gpu_cache(request):
    """After every -m gpu test: hand the device memory PyTorch's caching allocator still holds back to the driver.  The
    full-size tests leave up to 128 GiB cached in this process, and the multi-rank tests that follow run in SUBPROCESSES
    (ranks as threads or processes sharing the card) that need the memory themselves."""
    yield
    if request.node.get_closest_marker("gpu") is not None and "torch" in sys.modules:
        import torch
        if torch.cuda.is_available():
            import gc
            gc.collect()
            torch.cuda.empty_cache()
