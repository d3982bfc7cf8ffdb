import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _poison_free_gpu_memory(request):
    """Before every GPU test (AMX_TEST_POISON=0 switches it off) fill a few GB of the caching allocator's FREE memory with NaNs, so a kernel that
    reads (or relies on) memory nobody wrote cannot pass by luck on fresh zero pages.  On by default (costs ~0.1 s per test; it found a real arena overrun in round 1)."""
    if os.environ.get("AMX_TEST_POISON", "1") == "1" and request.node.get_closest_marker("gpu") is not None:
        import torch
        if torch.cuda.is_available():
            blocks = [torch.empty(64 << 20, dtype=torch.float32, device="cuda:0").fill_(float("nan")) for _ in range(8)]
            torch.cuda.synchronize()
            del blocks
    yield
