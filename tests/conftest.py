"""pytest configuration: marker registration and shared paths.

`-m "not gpu"` runs everything that needs no device (oracle vs golden vectors, host logic, C-ABI symbol
checks, gloo multi-rank tests); `-m gpu` runs the parity tests proper through the C-ABI on an MI355X.
"""
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture
def knobs():
    """The library's test switches (pie_set_knob; the environment is not read any more): knobs(name, value) sets one, all are back at
    their defaults after the test."""
    from proxy_inference_engine_amd import _ffi

    def setter(name, value):
        _ffi.set_knob(name, value)

    yield setter
    for name in _ffi.KNOBS:
        _ffi.set_knob(name, None)
