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


@pytest.fixture
def sweeps_always(monkeypatch):
    """In auto mode the library gives launches of a few frames to the pool path (both paths are exact); parity tests
    that mean to exercise the fused sweeps with small clips switch that off."""
    from avisynth_sangnom2_amd import capi
    monkeypatch.setitem(capi.POLICY_DEFAULTS, "small_launches", capi.SN_SMALL_SWEEP)
