"""`pie_core`: the reference's native module exports exactly hello() (src/pie_core/src/bindings.cpp:6-9)."""
from . import _ffi


def hello() -> str:
    return _ffi.hello()
