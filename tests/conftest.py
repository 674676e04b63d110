import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "small_launch_policy: keeps auto mode's pool path for small launches switched on")


@pytest.fixture(scope="session")
def hip_lib():
    """The in-tree libsangnom_hip.so, compiled first if a fresh checkout has not run __graft_entry__.build() yet
    (hipcc cross-compiles without a GPU).  Tests fail loudly, never skip, if it cannot be built or loaded."""
    from avisynth_sangnom2_amd import capi
    if not os.path.exists(capi.LIB_PATH):
        capi.build()
    return capi.load()


@pytest.fixture(scope="session", autouse=True)
def _device_to_host_through_pinned_memory():
    """The suite's own transfers never hand pageable memory to the HIP runtime: torch.from_numpy(...).pin_memory().to(dev)
    on the way in (the tests spell that out), and Tensor.cpu() -- patched here for the session -- through a pinned buffer
    on the way out.  One run of the GPU suite in twenty or so ended in a silent abort() inside a LATER test's pageable
    torch copy (twice at a `.cpu()` after the library's work had been waited for, once at a `.to(device)`; no library
    frame, no runtime message: DESIGN.md 7.6, profiles/r3_page_fault.md).  The library's own host entry points stage
    pageable planes through pinned buffers for the same reason."""
    try:
        import torch
    except ImportError:
        yield
        return
    if not torch.cuda.is_available():
        yield
        return
    plain_cpu = torch.Tensor.cpu

    def cpu(self, *args, **kwargs):
        if self.is_cuda and not args and not kwargs:
            buf = torch.empty(self.shape, dtype=self.dtype, pin_memory=True)
            buf.copy_(self)
            return buf
        return plain_cpu(self, *args, **kwargs)

    torch.Tensor.cpu = cpu
    yield
    torch.Tensor.cpu = plain_cpu


@pytest.fixture
def sweeps_always(monkeypatch):
    """In auto mode the library gives launches of a few frames to the pool path (both paths are exact); parity tests
    that mean to exercise the fused sweeps with small clips switch that off."""
    from avisynth_sangnom2_amd import capi
    monkeypatch.setitem(capi.POLICY_DEFAULTS, "small_launches", capi.SN_SMALL_SWEEP)
