import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _poison_torch_empty():
    """SNERF_TEST_POISON_EMPTY=1: every CUDA tensor from torch.empty / empty_like / new_empty starts as NaNs (floats), 0xff
    bytes (uint8 workspaces) or -1 (ints) - a read of memory the library has not written then shows up in the results
    instead of meeting whatever the caching allocator handed out.  For whole-suite runs by hand."""
    import torch

    def fill(t):
        if t.is_cuda and t.numel():
            if t.dtype.is_floating_point:
                t.fill_(float("nan"))
            elif t.dtype == torch.uint8:
                t.fill_(0xff)
            elif t.dtype in (torch.int32, torch.int64):
                t.fill_(-1)
        return t

    orig_empty, orig_like, orig_new = torch.empty, torch.empty_like, torch.Tensor.new_empty
    torch.empty = lambda *a, **k: fill(orig_empty(*a, **k))
    torch.empty_like = lambda *a, **k: fill(orig_like(*a, **k))
    torch.Tensor.new_empty = lambda self, *a, **k: fill(orig_new(self, *a, **k))


if os.environ.get("SNERF_TEST_POISON_EMPTY"):
    _poison_torch_empty()


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name)))


@pytest.fixture(scope="session")
def golden():
    return load_golden


@pytest.fixture(autouse=True)
def _no_leftover_checker_taps():
    """oracle/torch_cpu_path.FINE_OVERRIDE makes the CPU restatement reuse somebody else's hierarchical samples; it is set through
    the fine_override() context manager only, and no test may start or end with it set."""
    import sys
    mod = sys.modules.get("oracle.torch_cpu_path")
    assert mod is None or mod.FINE_OVERRIDE is None, "a checker tap was left set before this test"
    yield
    mod = sys.modules.get("oracle.torch_cpu_path")
    assert mod is None or mod.FINE_OVERRIDE is None, "this test left oracle.torch_cpu_path.FINE_OVERRIDE set"
